"""The split of a plane between AMaZE's two kernels (host logic, no GPU): k_amaze_rows.hip may only take tiles whose 160 rows and
columns lie inside the image and that no incomplete tile is chained behind (amaze_demosaic_RT.c's block of tile planes is never
cleared: an incomplete tile reads what the tile before it left there, mlvfs_amd/csrc/k_amaze.hip)."""
import ctypes as C
import pytest
from mlvfs_amd import lib


def split(L, w, h):
    a, b = C.c_int(-1), C.c_int(-1)
    L.mlvfs_amd_amaze_rows_extent(w, h, C.byref(a), C.byref(b))
    return a.value, b.value


def restated(w, h):
    step = 128
    tiles_x, tiles_y = (w + 16 + step - 1) // step, (h + 16 + step - 1) // step
    cc1_last, rr1_last = w + 16 - (-16 + (tiles_x - 1) * step), h + 16 - (-16 + (tiles_y - 1) * step)
    inc_x = 0 if cc1_last >= 160 else (2 if cc1_last < 32 else 1)
    inc_y = 0 if rr1_last >= 160 else (2 if rr1_last < 32 else 1)
    if tiles_x < inc_x + 1 or tiles_x < 3:
        return 0, 0                                      # small image: one workgroup walks it in the reference's order
    inside_x = [tx for tx in range(tiles_x) if -16 + tx * step + 160 <= w]
    inside_y = [ty for ty in range(tiles_y) if -16 + ty * step + 160 <= h]
    wgs_per_row, rows_a = tiles_x - inc_x, tiles_y - inc_y
    head = wgs_per_row - 1 if inc_x else None              # the tile an incomplete tile is chained behind stays with its chain
    fx = len([tx for tx in inside_x if tx < wgs_per_row and tx != head])
    fy = len([ty for ty in inside_y if ty < rows_a])
    return (fx, fy) if fx > 0 and fy > 0 else (0, 0)


def test_split_matches_the_restated_rule_for_every_geometry():
    L = lib.load()
    for w in list(range(36, 1200, 4)) + [1920, 2592, 3584, 4096, 5796]:
        for h in list(range(36, 700, 7)) + [540, 660, 1080, 1108, 1320, 2160]:
            assert split(L, w, h) == restated(w, h), (w, h)


@pytest.mark.parametrize("w,h,want", [(3584, 1320, (27, 10)), (3584, 660, (27, 5)), (1920, 1080, (14, 8)), (304, 304, (1, 2)), (260, 300, (1, 2)), (200, 300, (0, 0))])
def test_known_geometries(w, h, want):
    assert split(lib.load(), w, h) == want


def test_complete_tiles_are_a_prefix_of_the_grid():
    L = lib.load()
    for w, h in [(688, 560), (1332, 789), (3584, 1320)]:
        fx, fy = split(L, w, h)
        assert -16 + (fx - 1) * 128 + 160 <= w and -16 + (fy - 1) * 128 + 160 <= h


def test_heads_of_outputless_chains_only_where_the_width_is_a_multiple_of_128():
    """mlvfs_amd_amaze_rows_extra_mode's count: the tiles of column nfx in rows 0 .. n - 1 that k_amaze_rows.hip may take as well --
    only where the row's one incomplete tile is all apron (cc1 == 32), never the row whose chain feeds the incomplete bottom row."""
    L = lib.load()
    for w, h, want in [(3584, 1320, 9), (3584, 660, 4), (1920, 1080, 7), (3584, 1264, 8), (1332, 789, 0), (688, 560, 0), (384, 304, 1), (512, 400, 2)]:
        n = C.c_int(-1)
        before = L.mlvfs_amd_amaze_rows_extra_mode(-1, w, h, C.byref(n))
        assert before == -1 and n.value == want, (w, h, n.value)
        nfx, nfy = split(L, w, h)
        assert n.value <= nfy

"""k_amaze_rows.hip (AMaZE's complete tiles, row-streamed through LDS) against the oracle and against k_amaze.hip, plane by plane.

The two kernels implement mlvfs/amaze_demosaic_RT.c (SSE2 variant) with different schedules; `mlvfs_amd_amaze_debug` runs either
and copies the tile planes out.  The three output planes must equal the oracle's bit for bit whichever kernel takes the complete
tiles, and the planes a complete tile leaves behind must be the same in both kernels wherever the reference defines them."""
import ctypes as C
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

T, TT, HALF = 160, 160 * 160, 160 * 80
TILE = 13 * TT + 13 * HALF
FULL = ["cfa", "green", "delsq", "dw0", "dw1", "vcd", "hcd", "vcdalt", "hcdalt", "cdsq", "dgv", "dgh", "hcd2"]
HALFP = ["hvwt", "dgrb0", "dgrb1", "delp", "delm", "rbint", "curv_h", "curv_v", "sqm", "sqp", "pmwt", "rbm", "rbp"]


def _textured(w, h, seed):
    from mlvfs_amd import synth
    raw = synth.amaze_plane(w, h, seed)
    raw[::2, ::2] *= 1.0 + 0.5 * ((np.arange(w)[None, ::2] // 3) % 2)       # strong texture: the Nyquist test fires, the vote and the area pass run
    return raw.clip(0, 0xFFFFF).astype(np.float32)


def _run(gpu, raw, mode):
    import torch
    from mlvfs_amd import lib
    h, w = raw.shape
    d_raw = torch.from_numpy(raw).cuda()
    out = [torch.full((h, w), float("nan"), dtype=torch.float32, device="cuda") for _ in range(3)]
    tiles = ((w + 16 + 127) // 128) * ((h + 16 + 127) // 128)
    planes = torch.zeros(tiles * TILE, dtype=torch.float32, device="cuda")
    nfx, nfy = C.c_int(0), C.c_int(0)
    rc = gpu.mlvfs_amd_amaze_debug(C.c_void_p(d_raw.data_ptr()), w, h, *[C.c_void_p(t.data_ptr()) for t in out], mode,
                                   C.c_void_p(planes.data_ptr()), planes.numel(), C.byref(nfx), C.byref(nfy))
    assert rc == 0, lib.last_error()
    torch.cuda.synchronize()
    return [t.cpu().numpy() for t in out], planes.cpu().numpy(), nfx.value, nfy.value


def _plane(block, name):
    if name in FULL:
        k = FULL.index(name)
        return block[k * TT:(k + 1) * TT].reshape(T, T)
    k = HALFP.index(name)
    return block[13 * TT + k * HALF:13 * TT + (k + 1) * HALF].reshape(T, 80)


@pytest.fixture(scope="module")
def gpu():
    from mlvfs_amd import lib
    L = lib.load()
    assert L.mlvfs_amd_init(0) == 0
    return L


@pytest.fixture(scope="module")
def oracle():
    from oracle.bindings import Oracle
    return Oracle()


SIZES = [(304, 304), (432, 304), (560, 432), (688, 560), (1332, 789), (808, 1226), (1920, 540), (3584, 660)]


@pytest.mark.parametrize("w,h", SIZES)
def test_complete_tiles_through_lds_equal_the_oracle(gpu, oracle, w, h):
    raw = _textured(w, h, w * 7 + h)
    want = oracle.amaze_demosaic(raw)
    got, _, nfx, nfy = _run(gpu, raw, 1)
    assert nfx > 0 and nfy > 0, "the geometry has complete tiles"
    for g, x in zip(got, want):
        assert np.array_equal(g.view(np.uint32), x.view(np.uint32))


def test_the_tile_planes_of_the_two_kernels_agree(gpu):
    """Every plane of every complete tile, in the region every consumer reads (rows / columns 12 ... 147), bit for bit."""
    w, h = 688, 560
    raw = _textured(w, h, 99)
    _, pl0, _, _ = _run(gpu, raw, 0)
    _, pl1, nfx, nfy = _run(gpu, raw, 1)
    tiles_x = (w + 16 + 127) // 128
    yy, xx = np.mgrid[0:T, 0:T]
    rb_sites = ((yy + xx) % 2 == 0)
    for ty in range(nfy):
        for tx in range(nfx):
            b0 = pl0[(ty * tiles_x + tx) * TILE:][:TILE]
            b1 = pl1[(ty * nfx + tx) * TILE:][:TILE]
            for name in FULL + HALFP:
                a, b = _plane(b0, name), _plane(b1, name)
                d = a.view(np.uint32) != b.view(np.uint32)
                if name == "green":
                    d &= rb_sites                              # the rows kernel keeps G at R/B sites only (G sites are cfa)
                if name == "hcd":
                    continue                                   # k_amaze.hip refines hcd in place, k_amaze_rows.hip into hcd2
                inner = d[12:148, 12:148] if name in FULL else d[12:148, 6:74]
                assert not inner.any(), f"tile ({ty},{tx}) plane {name}: {int(inner.sum())} values differ"


def test_a_workgroup_streams_through_many_tiles(gpu, oracle, monkeypatch):
    """MLVFS_AMD_AMAZE_ROWS_WGS is read once per process: this test only checks the default grid on a plane with more tiles than CUs."""
    w, h = 128 * 20 + 32 + 128, 128 * 16 + 32 + 128               # 21 x 17 = 357 complete tiles > 256 workgroups
    raw = _textured(w, h, 5)
    want = oracle.amaze_demosaic(raw)
    got, _, nfx, nfy = _run(gpu, raw, 1)
    assert nfx * nfy > 256
    for g, x in zip(got, want):
        assert np.array_equal(g.view(np.uint32), x.view(np.uint32))


@pytest.mark.parametrize("w,h", [(384, 304), (512, 400), (640, 688), (1920, 540), (3584, 660)])
def test_heads_of_outputless_chains_through_the_row_kernel(gpu, oracle, w, h):
    """Widths that are a multiple of 128: the one incomplete tile of a tile row is 32 columns of apron without an output pixel, and
    the tile it is chained behind may go through k_amaze_rows.hip (right apron mirrored in its loader) -- VERDICT r3 next #2, built
    for the case where no stale plane is involved; off by default because the batch is no faster with it (k_amaze_rows.hip)."""
    import torch
    from mlvfs_amd import lib
    raw = _textured(w, h, w + 3 * h)
    want = oracle.amaze_demosaic(raw)
    n = C.c_int(0)
    before = gpu.mlvfs_amd_amaze_rows_extra_mode(1, w, h, C.byref(n))
    try:
        assert n.value > 0, "the geometry has such tiles"
        d_raw = torch.from_numpy(raw).cuda()
        out = [torch.full((h, w), float("nan"), dtype=torch.float32, device="cuda") for _ in range(3)]
        for rep in range(2):                                   # (twice: the second run finds the blocks the first one left)
            assert gpu.mlvfs_amd_amaze_demosaic_dev(C.c_void_p(d_raw.data_ptr()), w, h, *[C.c_void_p(t.data_ptr()) for t in out], None) == 0, lib.last_error()
            torch.cuda.synchronize()
            for g, x in zip(out, want):
                assert np.array_equal(g.cpu().numpy().view(np.uint32), x.view(np.uint32))
    finally:
        gpu.mlvfs_amd_amaze_rows_extra_mode(before, w, h, None)

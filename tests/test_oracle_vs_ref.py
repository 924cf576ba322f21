"""The oracle (repo's C restatement) against the REFERENCE'S OWN CODE compiled by
oracle/Makefile into oracle/_ref (this container only: needs /root/reference).

The reference ships no tests or golden vectors for this path, so
this comparison is what pins the oracle.  Byte-identical outputs are required on
seeded normal frames, on adversarial frames (40 % of pixels at black+-4 where
raw2ev is INT_MIN, pixels below black, clipped pixels), on ragged/edge geometries
and on every method / mode flag of the stage.
"""
import ctypes as C

import numpy as np
import pytest

from mlvfs_amd import synth

BLACK, WHITE = synth.BLACK, synth.WHITE
SIZES = [(64, 48), (136, 72), (258, 130), (416, 264)]
FRAMES = [("normal", 1), ("normal", 9), ("adversarial", 7), ("dual_iso", 3)]


def frame(kind, seed, w, h):
    return getattr(synth, kind + "_frame")(w, h, seed=seed)


def test_tables(oracle, reference):
    for black in (0, 1, 2047, 2048, 8000, 16384):
        assert np.array_equal(oracle.raw2ev(black, 16384 + 0), reference.raw2ev(black, 16384))
    assert np.array_equal(oracle.ev2raw(), reference.ev2raw())
    r2e = oracle.raw2ev(2048)
    assert r2e[2047] == 0 and r2e[2048] == -2**31 and r2e[2049] == 0 and r2e[2050] == 32768
    e2r = oracle.ev2raw()
    assert e2r[10 * 32768 - 1] == 0 and e2r[10 * 32768] == 1 and e2r[-1] == 16383       # SURVEY.md L1 spot values


def test_glibc_rand_restatement(oracle):
    libc = C.CDLL(None)
    libc.srand(1)
    want = [libc.rand() for _ in range(4096)]
    assert want[0] == 1804289383
    assert list(oracle.glibc_rand(4096)) == want


@pytest.mark.parametrize("bpp", [8, 10, 12, 14, 16])
@pytest.mark.parametrize("w,h", SIZES[:3])
def test_unpack(oracle, reference, w, h, bpp):
    f = frame("normal", 1, w, h) & ((1 << bpp) - 1)
    p = synth.pack_bits(f, bpp)
    a, b = oracle.unpack(p, w, h, bpp), reference.unpack(p, w, h, bpp)
    assert np.array_equal(a, b) and np.array_equal(a.reshape(h, w), f)
    for off, size in [(2 * 1000, 4000), (2 * 4097, 2 * 999), (-64, 4096)]:
        fw = (max(off, 0) // 2) * bpp // 16
        assert np.array_equal(oracle.unpack(p[fw:], w, h, bpp, off, size), reference.unpack(p[fw:], w, h, bpp, off, size))


@pytest.mark.parametrize("method", [2, 3, 5])
@pytest.mark.parametrize("kind,seed", FRAMES)
@pytest.mark.parametrize("w,h", SIZES)
def test_chroma_smooth(oracle, reference, w, h, kind, seed, method):
    f = frame(kind, seed, w, h)
    assert np.array_equal(oracle.chroma_smooth(f, BLACK, method), reference.chroma_smooth(f, BLACK, method))


@pytest.mark.parametrize("aggressive", [0, 1])
@pytest.mark.parametrize("dual_iso", [0, 1])
@pytest.mark.parametrize("kind,seed", FRAMES[:3])
@pytest.mark.parametrize("w,h", SIZES)
def test_bad_pixels(oracle, reference, w, h, kind, seed, dual_iso, aggressive):
    f = frame(kind, seed, w, h)
    assert np.array_equal(oracle.fix_bad_pixels(f, BLACK, aggressive, dual_iso),
                          reference.fix_bad_pixels(f, BLACK, aggressive, dual_iso))


def test_bad_pixels_crop(oracle, reference):
    f = frame("normal", 1, 136, 72)
    pan = (13, 7)
    crop = ((pan[0] + 7) & ~7, pan[1] & ~1)                      # cs.c:224-225
    assert np.array_equal(oracle.fix_bad_pixels(f, BLACK, 0, 0, crop), reference.fix_bad_pixels(f, BLACK, 0, 0, pan))


@pytest.mark.parametrize("dual_iso", [0, 1])
def test_focus_pixels(oracle, reference, tmp_path, monkeypatch, dual_iso):
    w, h, camera = 136, 72, 0x80000331
    rng = np.random.default_rng(5)
    pts = [(int(rng.integers(0, w)), int(rng.integers(0, h))) for _ in range(300)]
    pts += [(0, 10), (1, 1), (w - 1, 30), (w - 2, h - 1), (50, 0), (51, h - 1), (3, 3), (w - 4, h - 4),
            (60, 30), (62, 30), (61, 30), (60, 32), (60, 30)]
    monkeypatch.chdir(tmp_path)
    (tmp_path / f"{camera:x}_{w}x{h}.fpm").write_text("".join(f"{x} \t {y}\n" for x, y in pts))
    f = frame("normal", 1, w, h)
    # note: the reference caches the map per (camera, width, height) and its
    # free_focus_pixel_maps() leaves a stale count behind (cs.c:403-418), so it is never
    # called here; both parametrisations use the same list
    want = reference.fix_focus_pixels(f, BLACK, dual_iso, camera, w, h)
    assert np.array_equal(oracle.apply_focus_pixels(f, BLACK, np.array(pts, np.int32), (0, 0), dual_iso), want)
    assert (want != f).any()


@pytest.mark.parametrize("kind,seed", FRAMES[:3])
@pytest.mark.parametrize("w,h", SIZES)
def test_stripes(oracle, reference, w, h, kind, seed):
    f = frame(kind, seed, w, h)
    a, b = oracle.stripes_compute(f, BLACK, WHITE), reference.stripes_compute(f, BLACK, WHITE)
    assert a[0] == b[0] and np.array_equal(a[1], b[1])
    co = np.array([65536, 65536, 65354, 65741, 65240, 65866, 65448, 65640], np.int32)
    assert np.array_equal(oracle.stripes_apply(f, BLACK, WHITE, 1, co), reference.stripes_apply(f, BLACK, WHITE, 1, co))
    assert np.array_equal(oracle.stripes_apply(f, BLACK, WHITE, 0, co), f)


def test_stripes_row_shards_sum_to_whole(oracle):
    f = frame("normal", 1, 256, 130)
    needed, co, hist, num = oracle.stripes_compute(f, BLACK, WHITE, want_hist=True)
    total = oracle.stripes_hist_rows(f, 0, 130, BLACK, WHITE)
    assert total == int(num.sum())
    rnd = (oracle.glibc_rand(2 * total) % 1024).astype(np.uint16)
    acc_h, acc_n, first = np.zeros(8 * 65536, np.int32), np.zeros(8, np.int32), 0
    for r0, r1 in [(0, 40), (40, 41), (41, 130)]:
        n = oracle.stripes_hist_rows(f, r0, r1, BLACK, WHITE)
        _, hh, nn = oracle.stripes_hist_rows(f, r0, r1, BLACK, WHITE, rnd[2 * first: 2 * (first + n) + 2])
        acc_h += hh; acc_n += nn; first += n
    assert np.array_equal(acc_h.reshape(8, 65536), hist) and np.array_equal(acc_n, num)


@pytest.mark.parametrize("w,h", SIZES)
def test_hdr_preview(oracle, reference, w, h):
    for kind, seed in FRAMES:
        f = frame(kind, seed, w, h)
        a, b = oracle.hdr_preview(f, BLACK, WHITE), reference.hdr_preview(f, BLACK, WHITE)
        assert a[0] == b[0] and a[2] == b[2] and np.array_equal(a[1], b[1])
    assert oracle.hdr_preview(frame("dual_iso", 3, w, h), BLACK, WHITE)[0] == 1


@pytest.mark.parametrize("w,h", [(64, 48), (136, 72), (258, 130)])
def test_pattern_noise(oracle, reference, w, h):
    for kind, seed in FRAMES[:3]:
        f = frame(kind, seed, w, h)
        assert np.array_equal(oracle.fix_pattern_noise(f, WHITE), reference.fix_pattern_noise(f, WHITE))


@pytest.mark.parametrize("flags", [1, 2, 3, 4, 5, 8, 9, 6, 12])
def test_pattern_noise_debug_views(oracle, reference, flags):
    """patternnoise.c:215-240, 363-379: debug_flags choose one direction (bit 0: rows) and a view (denoised / noise / mask; the
    first of several wins).  MLVFS passes 0; the views complete fix_pattern_noise's behaviour."""
    for (w, h) in [(64, 48), (136, 72)]:
        f = frame(FRAMES[0][0], FRAMES[0][1], w, h)
        f[5:9, 20:30] = WHITE                      # something for the mask
        a, b = oracle.fix_pattern_noise(f, WHITE, flags), reference.fix_pattern_noise(f, WHITE, flags)
        assert np.array_equal(a, b)
        assert (a != f).any()


def test_histogram_16bit_counters(oracle, reference):
    data = np.full(70000, 100, np.uint16)                  # one bin receives 70000 samples: the counter wraps
    data[:3000] = 50
    h = oracle.L.orc_hist_create
    h.restype = C.c_void_p
    hp = h(C.c_uint16(1000))
    oracle.L.orc_hist_add.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint16]
    oracle.L.orc_hist_add(hp, data.ctypes.data, data.size, 0)
    oracle.L.orc_hist_median.restype = C.c_uint16
    oracle.L.orc_hist_median.argtypes = [C.c_void_p]
    assert oracle.L.orc_hist_median(hp) == reference.hist_median(data, 0, 1000)


@pytest.mark.parametrize("cs,bad,stripes", [(0, 0, 0), (2, 0, 0), (5, 1, 1), (3, 2, 1)])
def test_process_frame_order(oracle, reference, cs, bad, stripes):
    """main.c:942-997: unpack -> bad pixels -> chroma smooth -> stripes (computed on the
    post-bad-pix, post-chroma-smooth first frame)."""
    w, h = 256, 130
    ca = cb = None
    for fr in range(2):
        p = synth.pack_bits(synth.normal_frame(w, h, frame=fr))
        a, ca = oracle.process_frame(p, w, h, BLACK, WHITE, cs, bad, stripes, correction=ca)
        b, cb = reference.process_frame(p, w, h, BLACK, WHITE, cs, bad, stripes, correction=cb)
        assert np.array_equal(a, b)
        if stripes:
            assert ca[0] == cb[0] and np.array_equal(ca[1], cb[1])


# ------------------------------------------------------------------ full dual-ISO (mean23)
@pytest.mark.parametrize("fullres,alias", [(1, 1), (1, 0), (0, 1), (0, 0)])
@pytest.mark.parametrize("w,h", [(64, 48), (136, 72), (416, 264)])
def test_cr2hdr20_mean23(oracle, reference, w, h, fullres, alias):
    """hdr.c:1932-1957 with --mean23.  Both implementations keep the 20-bit tables of the first
    white level they see per black level; the synthetic dual-ISO frames share one white level,
    so the order of the calls does not matter here."""
    f = synth.dual_iso_frame(w, h)
    a = oracle.cr2hdr20(f, BLACK, WHITE, 1, fullres, alias, 0, reset=False)
    b = reference.cr2hdr20(f, BLACK, WHITE, 1, fullres, alias, 0)
    assert a[0] == b[0] == 1 and a[2] == b[2]
    assert np.array_equal(a[1], b[1])


def test_cr2hdr20_gbrg_and_rejects(oracle, reference):
    f = synth.dual_iso_frame(136, 74)[1:73].copy()
    a, b = oracle.cr2hdr20(f, BLACK, WHITE, 1, 1, 1, 0, reset=False), reference.cr2hdr20(f, BLACK, WHITE, 1, 1, 1, 0)
    assert a[0] == b[0] == 1 and np.array_equal(a[1], b[1])
    n = synth.normal_frame(136, 72)
    a, b = oracle.cr2hdr20(n, BLACK, WHITE, 1, 1, 1, 0, reset=False), reference.cr2hdr20(n, BLACK, WHITE, 1, 1, 1, 0)
    assert a[0] == b[0] == 0 and np.array_equal(a[1], n) and np.array_equal(b[1], n) and a[2] == b[2] == (BLACK, WHITE)


@pytest.mark.parametrize("fullres", [1, 0])
@pytest.mark.parametrize("cs", [2, 3, 5, 4])
@pytest.mark.parametrize("w,h", [(64, 48), (416, 264)])
def test_cr2hdr20_chroma_smooth(oracle, reference, w, h, cs, fullres):
    f = synth.dual_iso_frame(w, h)
    a = oracle.cr2hdr20(f, BLACK, WHITE, 1, fullres, 1, cs, reset=False)
    b = reference.cr2hdr20(f, BLACK, WHITE, 1, fullres, 1, cs)
    assert a[0] == b[0] == 1 and np.array_equal(a[1], b[1])


# ------------------------------------------------------------------ AMaZE (SSE2 variant) and the edge-directed dual-ISO path
@pytest.mark.parametrize("w,h", [(64, 48), (136, 72), (160, 160), (300, 200), (128, 130), (416, 264)])
def test_amaze_planes_bit_identical(oracle, reference, w, h):
    """amaze_demosaic_RT.c:113-1487 as built on x86-64 (SSE2 passes): all three float planes, bit for bit."""
    raw = synth.amaze_plane(w, h)
    for a, b in zip(oracle.amaze_demosaic(raw), reference.amaze_demosaic(raw)):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


@pytest.mark.parametrize("fullres,alias,cs", [(1, 1, 0), (0, 1, 0), (1, 0, 0), (1, 1, 5)])
@pytest.mark.parametrize("w,h", [(64, 48), (136, 72), (416, 264), (640, 400)])
def test_cr2hdr20_amaze_edge(oracle, reference, w, h, fullres, alias, cs):
    """BASELINE.json config 4 (--amaze-edge): hdr.c:954-1229 on top of the demosaic."""
    f = synth.dual_iso_frame(w, h)
    a = oracle.cr2hdr20(f, BLACK, WHITE, 0, fullres, alias, cs, reset=False)
    b = reference.cr2hdr20(f, BLACK, WHITE, 0, fullres, alias, cs)
    assert a[0] == b[0] == 1 and a[2] == b[2]
    assert np.array_equal(a[1], b[1])


def test_cr2hdr20_amaze_gbrg(oracle, reference):
    f = synth.dual_iso_frame(136, 74)[1:73].copy()
    a, b = oracle.cr2hdr20(f, BLACK, WHITE, 0, 1, 1, 0, reset=False), reference.cr2hdr20(f, BLACK, WHITE, 0, 1, 1, 0)
    assert a[0] == b[0] == 1 and np.array_equal(a[1], b[1])


# ------------------------------------------------------------------ the caller's table builders (main.c:128-196)
@pytest.mark.parametrize("black", [0, 1, 2048, 2047, 8000, 16384])
def test_restated_ev_tables_equal_the_reference_text(reference, black):
    """oracle/ref_luts.c (restatement) against get_raw2ev / get_raw2evf / get_ev2raw as main.c defines them -- the reference's
    own text, sliced out of main.c into the reference build by oracle/Makefile (main.c as a whole needs <fuse.h>)."""
    from oracle import bindings
    want = reference.ev_tables(black)
    got = bindings.restated_ev_tables(black)
    for a, b, name in zip(got, want, ("raw2ev", "raw2evf", "ev2raw")):
        assert a.shape == b.shape and np.array_equal(a.view(np.uint8), b.view(np.uint8)), name       # doubles: bit for bit (-inf included)
    assert want[0][black] == -2 ** 31 and (want[0][:black] == 0).all()                                # the semantics the GPU tables bake in


def test_library_host_tables_equal_the_reference_text(reference, amd):
    """The library's own formulas (csrc/runtime.cpp, used when the caller provides no get_raw2ev): the self test re-derives the
    16-bit re-encodings from them; here the formulas themselves against the reference's tables, through the exported check."""
    r, rf, e = reference.ev_tables(0)
    lin = np.ascontiguousarray(r[:16384].astype(np.int32))
    e2r = np.ascontiguousarray(e.astype(np.int32))
    assert amd.mlvfs_amd_selftest_tables(lin.ctypes.data, e2r.ctypes.data) == 0

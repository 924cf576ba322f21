"""GPU parity: libmlvfs_amd.so (HIP, through the C ABI) against the oracle.

Bit-exact is the bar for every stage here (integer / byte work).  Where the
reference build (oracle/_ref) travelled to the GPU box it is checked too.
All tests are @pytest.mark.gpu and call through the C ABI: the drop-in symbols
the way MLVFS's main.c calls them (mlvfs_amd.pipeline) and the device-resident
API (mlvfs_amd.stream).
"""
import ctypes as C
import os

import numpy as np
import pytest

from mlvfs_amd import abi, lib, pipeline, synth

pytestmark = pytest.mark.gpu

BLACK, WHITE = synth.BLACK, synth.WHITE
SIZES = [(64, 48), (136, 72), (256, 130), (416, 264)]          # 136: not a multiple of 16 -> generic loaders
KINDS = ["normal", "adversarial"]


def frame_of(kind, w, h, **kw):
    return getattr(synth, kind + "_frame")(w, h, **kw)


def fh_for(w, h, bpp=14, **kw):
    return abi.make_frame_headers(w, h, bpp=bpp, black=BLACK, white=WHITE, **kw)


# ------------------------------------------------------------------ unpack
@pytest.mark.parametrize("bpp", [14, 12, 10, 8, 16])
@pytest.mark.parametrize("w,h", [(64, 48), (136, 72), (1920, 1080)])
def test_unpack_dropin(gpu, oracle, w, h, bpp):
    f = frame_of("normal", w, h) & ((1 << bpp) - 1)
    packed = synth.pack_bits(f, bpp)
    fh = fh_for(w, h, bpp)
    got = pipeline.get_image_data(fh, packed).reshape(h, w)
    assert np.array_equal(got, f)
    assert np.array_equal(got.ravel(), oracle.unpack(packed, w, h, bpp))


@pytest.mark.parametrize("offset,size", [(0, 4096), (2 * 1000, 6000), (2 * 4097, 2 * 999), (-64, 4096)])
def test_unpack_window(gpu, oracle, offset, size):
    """offset/max_size window of dng_get_image_data (dng.c:815-826); the caller hands in
    the packed words starting at the first requested pixel's word (main.c:689)."""
    w, h, bpp = 256, 130, 14
    f = frame_of("normal", w, h)
    packed = synth.pack_bits(f, bpp)
    first_word = (max(offset, 0) // 2) * bpp // 16
    want = oracle.unpack(packed[first_word:], w, h, bpp, offset=offset, max_size=size)
    fh = fh_for(w, h, bpp)
    got = pipeline.get_image_data(fh, packed[first_word:], offset=offset, max_size=size)
    assert np.array_equal(got, want)


# ------------------------------------------------------------------ chroma smooth
@pytest.mark.parametrize("method", [2, 3, 5])
@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("w,h", SIZES)
def test_chroma_smooth_dropin(gpu, oracle, w, h, kind, method):
    f = frame_of(kind, w, h)
    want = oracle.chroma_smooth(f, BLACK, method)
    got = f.copy()
    gpu.chroma_smooth(C.byref(fh_for(w, h)), lib.ptr(got), method)
    assert np.array_equal(got, want), f"{(got != want).sum()} px differ"
    assert (want != f).any()


@pytest.mark.parametrize("method", [2, 3, 5])
@pytest.mark.parametrize("w,h", [(64, 48), (416, 264), (3584, 1320)])
def test_chroma_smooth_colour_cast(gpu, oracle, w, h, method):
    """Footage-like colour balance (R, B well below G) with hard colour edges: the packed 16-bit medians of the 5x5
    kernel work relative to a local reference and fall back to 32 bits across the edges; both must be exact."""
    f = synth.colour_cast_frame(w, h)
    want = oracle.chroma_smooth(f, BLACK, method)
    got = f.copy()
    gpu.chroma_smooth(C.byref(fh_for(w, h)), lib.ptr(got), method)
    assert np.array_equal(got, want), f"{(got != want).sum()} px differ"
    assert (want != f).mean() > 0.2


def test_chroma_smooth_bad_method_is_noop(gpu):
    f = frame_of("normal", 64, 48)
    got = f.copy()
    gpu.chroma_smooth(C.byref(fh_for(64, 48)), lib.ptr(got), 4)
    assert np.array_equal(got, f)


def test_chroma_smooth_other_black_levels(gpu, oracle):
    """Every black level has an output table of its own (k_frame.hip: E2D_RECORDS, built and checked on the device when the level is
    first seen); 8192 is what dual ISO makes of 2048 (the conversion multiplies the levels by four)."""
    for black in (0, 1, 1024, 4000, 8192, 12000):
        f = synth.adversarial_frame(128, 64, black=black)
        fh = abi.make_frame_headers(128, 64, black=black, white=WHITE)
        for method in (2, 3, 5):
            got = f.copy()
            gpu.chroma_smooth(C.byref(fh), lib.ptr(got), method)
            assert np.array_equal(got, oracle.chroma_smooth(f, black, method)), (black, method)


# ------------------------------------------------------------------ bad / focus pixels
@pytest.mark.parametrize("aggressive", [0, 1])
@pytest.mark.parametrize("dual_iso", [0, 1])
@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("w,h", SIZES[:3])
def test_fix_bad_pixels_dropin(gpu, oracle, w, h, kind, dual_iso, aggressive):
    f = frame_of(kind, w, h)
    want = oracle.fix_bad_pixels(f, BLACK, aggressive, dual_iso)
    got = f.copy()
    gpu.fix_bad_pixels(C.byref(fh_for(w, h)), lib.ptr(got), aggressive, dual_iso)      # guid 0: detect every call
    assert np.array_equal(got, want), f"{(got != want).sum()} px differ"
    assert (want != f).any()


def test_fix_bad_pixels_map_cache_and_crop(gpu, oracle):
    """A clip (fileGuid != 0) detects once; later frames reuse the map (cs.c:233-253).  panPos
    shifts the stored coordinates and is subtracted again on application."""
    w, h = 256, 130
    f0, f1 = synth.normal_frame(w, h, frame=0), synth.normal_frame(w, h, frame=1)
    fh = fh_for(w, h, guid=0xABCDEF01, pan=(13, 7))
    crop = ((13 + 7) & ~7, 7 & ~1)
    pixels = oracle.detect_bad_pixels(f0, BLACK, 0, crop)
    got0, got1 = f0.copy(), f1.copy()
    gpu.fix_bad_pixels(C.byref(fh), lib.ptr(got0), 0, 0)
    gpu.fix_bad_pixels(C.byref(fh), lib.ptr(got1), 0, 0)            # uses frame 0's map
    assert np.array_equal(got0, oracle.apply_bad_pixels(f0, BLACK, pixels, crop))
    assert np.array_equal(got1, oracle.apply_bad_pixels(f1, BLACK, pixels, crop))
    gpu.free_focus_pixel_maps()


@pytest.mark.parametrize("dual_iso", [0, 1])
def test_fix_focus_pixels_dropin(gpu, oracle, tmp_path, monkeypatch, dual_iso):
    """Focus maps are read from '<camera hex>_<w>x<h>.fpm' in the CWD (cs.c:369-370);
    the list is applied in file order, with the frame-edge rules of cs.c:479-500."""
    w, h = 136, 72
    rng = np.random.default_rng(5)
    pts = [(rng.integers(0, w), rng.integers(0, h)) for _ in range(300)]
    pts += [(0, 10), (1, 1), (w - 1, 30), (w - 2, h - 1), (50, 0), (51, h - 1), (3, 3), (w - 4, h - 4),
            (60, 30), (62, 30), (61, 30), (60, 32), (60, 30)]          # dependent + duplicate entries
    camera = 0x80000331
    monkeypatch.chdir(tmp_path)
    (tmp_path / f"{camera:x}_{w}x{h}.fpm").write_text("".join(f"{x} \t {y}\n" for x, y in pts))
    gpu.free_focus_pixel_maps()
    f = synth.normal_frame(w, h)
    want = oracle.apply_focus_pixels(f, BLACK, np.array(pts, np.int32), (0, 0), dual_iso)
    got = f.copy()
    gpu.fix_focus_pixels(C.byref(fh_for(w, h, camera=camera)), lib.ptr(got), dual_iso)
    assert np.array_equal(got, want), f"{(got != want).sum()} px differ"
    assert (want != f).any()
    gpu.free_focus_pixel_maps()


# ------------------------------------------------------------------ stripes
@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("w,h", [(64, 48), (256, 130), (416, 264)])
def test_stripes_dropin(gpu, oracle, w, h, kind):
    f = frame_of(kind, w, h)
    needed, coeffs = oracle.stripes_compute(f, BLACK, WHITE)          # srand(1) + libc rand()
    libc = C.CDLL(None)
    libc.srand(1)
    name = f"clip_{kind}_{w}x{h}.MLV".encode()
    corr = gpu.stripes_new_correction(name)
    fh = fh_for(w, h)
    gpu.stripes_compute_correction(C.byref(fh), corr, lib.ptr(f), 0, f.size)
    assert corr.contents.correction_needed == needed
    assert list(corr.contents.coeffficients) == list(coeffs)
    # the libc generator must have been advanced by exactly the reference's number of calls
    after_gpu = libc.rand()
    libc.srand(1)
    oracle.stripes_compute(f, BLACK, WHITE, reseed=False)
    assert libc.rand() == after_gpu
    got = f.copy()
    gpu.stripes_apply_correction(C.byref(fh), corr, lib.ptr(got), 0, got.size)
    assert np.array_equal(got, oracle.stripes_apply(f, BLACK, WHITE, needed, coeffs))
    assert gpu.stripes_get_correction(name)
    gpu.stripes_free_corrections()
    assert not gpu.stripes_get_correction(name)


def test_stripes_dropin_with_the_applications_own_rand_state(gpu, oracle):
    """The dither comes from the application's libc generator.  Its default (TYPE_3: what srand() / rand() use) is advanced in bulk from
    its parked state (runtime.cpp: take_app_state / put_app_state); an application that installed a generator of another size with
    initstate() gets its values call by call.  Either way: the reference's coefficients and the reference's position in the stream."""
    libc = C.CDLL(None)
    libc.initstate.restype = C.c_void_p
    libc.setstate.restype = C.c_void_p
    libc.setstate.argtypes = [C.c_void_p]
    f = frame_of(KINDS[0], 416, 264)
    fh = fh_for(416, 264)
    for size, seed in ((256, 7), (8, 9), (128, 11), (64, 5)):            # TYPE_4, TYPE_0, TYPE_3 in a buffer of the application's, TYPE_2
        want_buf, got_buf = C.create_string_buffer(size), C.create_string_buffer(size)
        old = libc.initstate(seed, want_buf, size)
        needed, coeffs = oracle.stripes_compute(f, BLACK, WHITE, reseed=False)
        want_next = libc.rand()
        libc.initstate(seed, got_buf, size)
        corr = gpu.stripes_new_correction(f"own_state_{size}.MLV".encode())
        gpu.stripes_compute_correction(C.byref(fh), corr, lib.ptr(f), 0, f.size)
        got_next = libc.rand()
        libc.setstate(old)                                               # the process's default generator back
        assert corr.contents.correction_needed == needed and list(corr.contents.coeffficients) == list(coeffs), size
        assert got_next == want_next, size
    # a srand() in the middle of a run of calls: the bulk path must pick the state up wherever it is
    for seed, burn in ((3, 0), (3, 17), (12345, 1000)):
        libc.srand(seed)
        for _ in range(burn):
            libc.rand()
        needed, coeffs = oracle.stripes_compute(f, BLACK, WHITE, reseed=False)
        want_next = [libc.rand() for _ in range(40)]
        libc.srand(seed)
        for _ in range(burn):
            libc.rand()
        corr = gpu.stripes_new_correction(f"burn_{seed}_{burn}.MLV".encode())
        gpu.stripes_compute_correction(C.byref(fh), corr, lib.ptr(f), 0, f.size)
        assert [libc.rand() for _ in range(40)] == want_next
        assert corr.contents.correction_needed == needed and list(corr.contents.coeffficients) == list(coeffs)
    gpu.stripes_free_corrections()


def test_stripes_apply_noop_cases(gpu):
    w, h = 68, 16                                                       # xRes % 8 != 0 -> untouched (stripes.c:253)
    f = synth.normal_frame(w, h)
    corr = gpu.stripes_new_correction(b"x.MLV")
    corr.contents.correction_needed = 1
    for k in range(8):
        corr.contents.coeffficients[k] = 70000
    got = f.copy()
    gpu.stripes_apply_correction(C.byref(fh_for(w, h)), corr, lib.ptr(got), 0, got.size)
    assert np.array_equal(got, f)
    gpu.stripes_free_corrections()


# ------------------------------------------------------------------ dual-ISO preview
@pytest.mark.parametrize("w,h", [(64, 48), (136, 72), (416, 264)])
def test_hdr_preview_dropin(gpu, oracle, w, h):
    f = synth.dual_iso_frame(w, h)
    ok, want, levels = oracle.hdr_preview(f, BLACK, WHITE)
    assert ok == 1
    fh = fh_for(w, h)
    got = f.copy()
    r = gpu.hdr_convert_data(C.byref(fh), lib.ptr(got), 0, got.nbytes)
    assert r == 1
    assert np.array_equal(got, want), f"{(got != want).sum()} px differ"
    assert (fh.rawi_hdr.raw_info.black_level, fh.rawi_hdr.raw_info.white_level) == levels
    # a normal frame is not dual ISO: untouched, returns 0 (caller then runs the normal path)
    n = synth.normal_frame(w, h)
    got = n.copy()
    fh = fh_for(w, h)
    assert gpu.hdr_convert_data(C.byref(fh), lib.ptr(got), 0, got.nbytes) == 0
    assert np.array_equal(got, n) and fh.rawi_hdr.raw_info.black_level == BLACK


@pytest.mark.parametrize("w,h", [(416, 264), (258, 131)])
def test_hdr_preview_chains_of_rewritten_rows(gpu, oracle, w, h):
    """hdr.c:178-215 rewrites rows top-down in place: a clipped bright pixel and a dark pixel in deep shadow take the REWRITTEN row
    two above.  Columns where such pixels follow each other for the whole height (and random mixtures, and the first / last rows)
    make chains of any length; the per-pixel kernel must follow them like the reference's walk does."""
    import torch
    rng = np.random.default_rng(7)
    f = synth.dual_iso_frame(w, h)
    bright = (np.arange(h) % 4 >= 2)[:, None] & np.ones((1, w), bool)
    special = np.where(bright, 16383, BLACK - 40).astype(np.uint16)
    cols = np.zeros((h, w), bool)
    cols[:, 10:w:9] = True                                               # whole columns
    cols[:, 14:w:9] = rng.random((h, len(range(14, w, 9)))) < 0.6        # mixtures: chains of every length
    cols[:, 17:w:9] = (np.arange(h) % 8 < 6)[:, None]                    # runs of three pairs
    f = np.where(cols, special, f).astype(np.uint16)
    ok, want, levels = oracle.hdr_preview(f, BLACK, WHITE)
    assert ok == 1
    fh = fh_for(w, h)
    got = f.copy()
    assert gpu.hdr_convert_data(C.byref(fh), lib.ptr(got), 0, got.nbytes) == 1
    assert np.array_equal(got, want), f"{(got != want).sum()} px differ"
    # the device-resident entry rewrites the caller's buffer
    geom = lib.Geom(w, h, 14, BLACK, WHITE, 0, 0)
    t = torch.from_numpy(f.view(np.int16)).cuda()
    assert gpu.mlvfs_amd_hdr_preview_dev(C.byref(geom), C.c_void_p(t.data_ptr()), t.numel() * 2, None) == 1
    torch.cuda.synchronize()
    assert np.array_equal(t.cpu().numpy().view(np.uint16), want)


def test_hdr_preview_with_focus_map(gpu, oracle, tmp_path, monkeypatch):
    """hdr.c:104: the preview repairs the camera's focus pixels (dual-ISO rule) on the host frame before it matches the
    exposures -- a drop-in symbol calling another one, in every MLVFS_AMD_RESIDENT mode."""
    w, h = 416, 264
    ys, xs = np.mgrid[8:h - 8:6, 9:w - 9:8]
    pts = np.stack([xs.reshape(-1), ys.reshape(-1)], 1).astype(np.int32)
    camera = 0x80000331
    monkeypatch.chdir(tmp_path)
    (tmp_path / f"{camera:x}_{w}x{h}.fpm").write_text("".join(f"{x} \t {y}\n" for x, y in pts))
    gpu.free_focus_pixel_maps()
    f = synth.dual_iso_frame(w, h)
    fixed = oracle.apply_focus_pixels(f, BLACK, pts, (0, 0), 1)
    assert (fixed != f).any()
    ok, want, levels = oracle.hdr_preview(fixed, BLACK, WHITE)
    assert ok == 1
    fh = fh_for(w, h, camera=camera)
    got = f.copy()
    assert gpu.hdr_convert_data(C.byref(fh), lib.ptr(got), 0, got.nbytes) == 1
    assert np.array_equal(got, want), f"{(got != want).sum()} px differ"
    gpu.free_focus_pixel_maps()


# ------------------------------------------------------------------ process_frame order through the drop-in symbols
@pytest.mark.parametrize("cs,bad,stripes", [(0, 0, 0), (2, 0, 0), (5, 1, 1), (3, 2, 1), (5, 0, 1)])
def test_process_frame_dropin(gpu, oracle, cs, bad, stripes):
    w, h = 256, 130
    opt = pipeline.MlvfsOptions(chroma_smooth=cs, fix_bad_pixels=bad, fix_stripes=stripes)
    libc = C.CDLL(None)
    corr = None
    gpu.stripes_free_corrections()
    gpu.free_focus_pixel_maps()
    for fr in range(2):                                                # frame 0 computes the clip state, frame 1 reuses it
        f = synth.normal_frame(w, h, frame=fr)
        packed = synth.pack_bits(f)
        if fr == 0:
            libc.srand(1)
        want, corr = oracle.process_frame(packed, w, h, BLACK, WHITE, cs=cs, bad_pix=bad, stripes=stripes, correction=corr)
        if fr == 0:
            libc.srand(1)
        got = pipeline.process_frame(packed, fh_for(w, h), opt, "a.MLV")     # guid 0: bad pixels re-detected per frame, like the oracle
        assert np.array_equal(got, want), f"frame {fr}: {(got != want).sum()} px differ"
    gpu.stripes_free_corrections()


def test_against_reference_build(gpu, reference):
    """Where oracle/_ref travelled to this box: HIP == the reference's own code."""
    w, h = 256, 130
    for kind in KINDS:
        f = frame_of(kind, w, h)
        for m in (2, 3, 5):
            got = f.copy()
            gpu.chroma_smooth(C.byref(fh_for(w, h)), lib.ptr(got), m)
            assert np.array_equal(got, reference.chroma_smooth(f, BLACK, m))
        got = f.copy()
        gpu.fix_bad_pixels(C.byref(fh_for(w, h)), lib.ptr(got), 1, 0)
        assert np.array_equal(got, reference.fix_bad_pixels(f, BLACK, 1, 0))


# ------------------------------------------------------------------ pattern noise
@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("w,h", [(64, 48), (136, 72), (258, 130), (416, 264), (70, 50), (54, 38), (34, 44), (1002, 46), (46, 1002)])
def test_fix_pattern_noise_dropin(gpu, oracle, w, h, kind):
    """patternnoise.c:357-380: column pass, then the same pass on the transposed frame;
    int16 arithmetic, lower medians -- bit-exact."""
    f = frame_of(kind, w, h)
    want = oracle.fix_pattern_noise(f, WHITE)
    got = f.copy()
    gpu.fix_pattern_noise(lib.ptr(got), w, h, WHITE, 0)
    assert np.array_equal(got, want), f"{(got != want).sum()} px differ"
    assert (want != f).any()


@pytest.mark.parametrize("flags", [1, 2, 3, 4, 5, 8, 9, 6, 12])
def test_fix_pattern_noise_debug_views(gpu, oracle, flags):
    """patternnoise.c:215-240, 363-379 (debug_flags != 0; MLVFS passes 0): one direction only, and the denoised / noise / mask
    view of that direction's pass instead of its correction."""
    for (w, h) in [(64, 48), (136, 72), (416, 264), (46, 1002)]:
        f = frame_of(KINDS[0], w, h)
        f[5:9, 20:30] = WHITE
        want = oracle.fix_pattern_noise(f, WHITE, flags)
        got = f.copy()
        gpu.fix_pattern_noise(lib.ptr(got), w, h, WHITE, flags)
        assert np.array_equal(got, want), f"{w}x{h} flags {flags}: {(got != want).sum()} px differ"


def test_fix_pattern_noise_full_size_hash(gpu):
    """1920x1080 frame against the hash of the reference's output (tests/golden/golden.json)."""
    import json
    from conftest import fnv1a
    meta = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "golden.json")))["full_size"]
    if "A_1920x1080_pattern_noise" not in meta:
        pytest.skip("golden.json predates the pattern-noise hash")
    f = synth.normal_frame(1920, 1080, seed=1)
    got = f.copy()
    gpu.fix_pattern_noise(lib.ptr(got), 1920, 1080, WHITE, 0)
    assert fnv1a(got) == meta["A_1920x1080_pattern_noise"]


# ------------------------------------------------------------------ full size (3584x1320) against the reference's hashes
def _golden_full():
    import json
    return json.load(open(os.path.join(os.path.dirname(__file__), "golden", "golden.json")))["full_size"]


def test_fix_pattern_noise_benchmark_size_hash(gpu):
    """P1 at 3584x1320 (BASELINE configs' geometry) against the hash of the reference's output."""
    from conftest import fnv1a
    w, h = 3584, 1320
    got = synth.normal_frame(w, h, seed=1)
    gpu.fix_pattern_noise(lib.ptr(got), w, h, WHITE, 0)
    assert fnv1a(got) == _golden_full()["B_3584x1320_pattern_noise"]


def test_hdr_preview_benchmark_size_hash(gpu):
    """H1 hdr_convert_data at 3584x1320: pixels and the levels it writes back (hdr.c:223-224) against the reference's."""
    from conftest import fnv1a
    w, h = 3584, 1320
    got = synth.dual_iso_frame(w, h)
    fh = fh_for(w, h)
    assert gpu.hdr_convert_data(C.byref(fh), lib.ptr(got), 0, got.nbytes) == 1
    full = _golden_full()
    assert fnv1a(got) == full["B_3584x1320_hdr_preview"]
    assert [fh.rawi_hdr.raw_info.black_level, fh.rawi_hdr.raw_info.white_level] == full["B_3584x1320_hdr_preview_levels"]


def test_chroma_smooth_3x3_benchmark_frames_hash(gpu):
    """cs3x3 on the benchmark's frames 0 and 1 through the drop-in symbol against the reference's hashes."""
    from conftest import fnv1a
    w, h = 3584, 1320
    full = _golden_full()
    for k in range(2):
        got = synth.normal_frame(w, h, seed=1, frame=k)
        gpu.chroma_smooth(C.byref(fh_for(w, h)), lib.ptr(got), 3)
        assert fnv1a(got) == full[f"B_cs3_frame{k}"], k

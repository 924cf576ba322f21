"""GPU parity of the device-resident API and the fused pipeline (mlvfs_amd.stream),
plus size-independent properties at BASELINE.json's full frame size.

Everything is bit-exact (u16 / int32 work).
"""
import ctypes as C

import os
import numpy as np
import pytest

from mlvfs_amd import lib, synth

pytestmark = pytest.mark.gpu

BLACK, WHITE = synth.BLACK, synth.WHITE


@pytest.fixture(scope="module")
def torch_cuda(gpu):
    import torch
    assert torch.cuda.is_available()
    return torch


def make_stream(w, h):
    from mlvfs_amd.stream import ClipStream
    return ClipStream(w, h, 14, BLACK, WHITE, device=0)


def oracle_clip(oracle, frames, w, h, cs, bad, stripes):
    """Reference semantics for a clip: frame 0 fixes the pixel map and the stripe
    coefficients (main.c:969-988), later frames reuse them."""
    out, pixels, corr = [], None, None
    for k, f in enumerate(frames):
        img = f.copy()
        if bad:
            if pixels is None:
                pixels = oracle.detect_bad_pixels(img, BLACK, int(bad == 2))
            img = oracle.apply_bad_pixels(img, BLACK, pixels)
        if cs:
            img = oracle.chroma_smooth(img, BLACK, cs)
        if stripes:
            if corr is None:
                corr = oracle.stripes_compute(img, BLACK, WHITE, frame_size=w * h * 14 // 8)
            img = oracle.stripes_apply(img, BLACK, WHITE, *corr)
        out.append(img)
    return out, pixels, corr


@pytest.mark.parametrize("w,h", [(256, 130), (136, 72), (416, 264)])
@pytest.mark.parametrize("cs,bad,stripes", [(5, 1, 1), (2, 0, 0), (3, 2, 1), (0, 1, 1), (0, 0, 0)])
def test_fused_pipeline_matches_oracle(torch_cuda, oracle, w, h, cs, bad, stripes):
    from mlvfs_amd.stream import to_numpy_u16
    frames = [synth.normal_frame(w, h, frame=k) for k in range(3)]
    want, pixels, corr = oracle_clip(oracle, frames, w, h, cs, bad, stripes)
    s = make_stream(w, h)
    packed = s.upload_packed([synth.pack_bits(f) for f in frames])
    first = s.analyse_first_frame(packed, cs=cs, bad_pix=bad, stripes=bool(stripes), rand_mode=1)
    assert np.array_equal(to_numpy_u16(first)[0], want[0]), "first-frame path"
    if bad:
        assert np.array_equal(s.get_pixel_map(), pixels)
    if stripes:
        needed, co = s.get_stripes()
        assert needed == corr[0] and list(co) == list(corr[1])
    got = to_numpy_u16(s.process(packed, cs=cs, fix_pixels=bool(bad), stripes=bool(stripes)))
    for k in range(3):
        assert np.array_equal(got[k], want[k]), f"frame {k}: {(got[k] != want[k]).sum()} px differ"
    s.close()


@pytest.mark.parametrize("w,h", [(136, 62), (392, 58), (264, 88), (152, 46), (1736, 64), (1880, 34)])
@pytest.mark.parametrize("cs", [0, 2, 3, 5])
def test_fused_pipeline_rows_of_half_groups(torch_cuda, oracle, w, h, cs):
    """Widths that are a multiple of 8 but not of 16 (1736, 1880: real crop modes): every other row starts in the middle of a
    16-pixel group of the 14-bit stream, the last group of a row has eight pixels in the frame, and where w = 128 k + 8 that
    half group is also a tile's right halo -- on the frame's last row the one place where a 16-pixel read would leave the frame.
    Packed and 16-bit input, heights that end inside a tile, frames with pixels at black."""
    from mlvfs_amd.stream import to_numpy_u16
    assert w % 16 == 8
    for kind in ("normal", "low_light"):
        gen = getattr(synth, kind + "_frame")
        frames = [gen(w, h, seed=5 + k) if kind == "low_light" else gen(w, h, frame=k) for k in range(3)]
        want, pixels, corr = oracle_clip(oracle, frames, w, h, cs, 1, 1)
        s = make_stream(w, h)
        packed = s.upload_packed([synth.pack_bits(f) for f in frames])
        s.analyse_first_frame(packed, cs=cs, bad_pix=1, stripes=True, rand_mode=1)
        got = to_numpy_u16(s.process(packed, cs=cs, fix_pixels=True, stripes=True))
        got16 = to_numpy_u16(s.process_unpacked(s.unpack(packed), cs=cs, fix_pixels=True, stripes=True))
        for k in range(3):
            assert np.array_equal(got[k], want[k]), f"{kind} frame {k}: {(got[k] != want[k]).sum()} px differ"
            assert np.array_equal(got16[k], want[k]), f"{kind} frame {k}, 16-bit input: {(got16[k] != want[k]).sum()} px differ"
        s.close()


@pytest.mark.parametrize("w,h", [(256, 130), (1008, 44), (16, 12), (48, 10), (496, 122), (512, 124), (1736, 64), (136, 62), (3584, 66), (608, 250), (656, 190), (3584, 252)])
@pytest.mark.parametrize("cs,stripes", [(2, 0), (2, 1), (3, 0), (3, 1)])
def test_streaming_kernel_matches_oracle(torch_cuda, oracle, w, h, cs, stripes, monkeypatch):
    """k_frame_s (cs2x2 / cs3x3 without a pixel map: a wave per 62-item column, rows in registers, no barriers) takes long launches only
    (and hands footage with many pixels at or below black back to k_frame); MLVFS_AMD_KF_S=2 forces it wherever it can run, so that
    its corners are compared with the oracle: one and several columns (496 px each), a last column of one item, widths that are 8
    mod 16, two-row frames, frames of one task and of several, every footage kind (the loader's form for pixels at or below
    black, clamped look-ups), with and without the stripes epilogue.  The same launches through k_frame (MLVFS_AMD_KF_S=0) must
    give the same bytes."""
    from mlvfs_amd.stream import to_numpy_u16
    for kind in ("normal", "low_light", "colour_cast", "adversarial"):
        if kind in ("low_light", "colour_cast"):
            frames = [getattr(synth, kind + "_frame")(w, h, seed=3 + k) for k in range(3)]
        else:
            frames = [getattr(synth, kind + "_frame")(w, h, frame=k) for k in range(3)]
        want, _, corr = oracle_clip(oracle, frames, w, h, cs, 0, stripes)
        out = {}
        for mode in ("2", "0"):
            monkeypatch.setenv("MLVFS_AMD_KF_S", mode)
            s = make_stream(w, h)
            packed = s.upload_packed([synth.pack_bits(f) for f in frames])
            s.analyse_first_frame(packed, cs=cs, bad_pix=0, stripes=bool(stripes), rand_mode=1)
            out[mode] = to_numpy_u16(s.process(packed, cs=cs, fix_pixels=False, stripes=bool(stripes)))
            s.close()
        for k in range(3):
            assert np.array_equal(out["2"][k], want[k]), f"{kind} frame {k}: {(out['2'][k] != want[k]).sum()} px differ (k_frame_s)"
            assert np.array_equal(out["0"][k], want[k]), f"{kind} frame {k}: k_frame"


@pytest.mark.parametrize("w,h", [(256, 130), (1008, 64), (16, 12), (496, 122), (512, 124), (1736, 64), (3584, 66), (656, 190), (608, 250)])
@pytest.mark.parametrize("bad,stripes", [(0, 0), (1, 1), (2, 0)])
def test_streaming_cs5x5_kernel_matches_oracle(torch_cuda, oracle, w, h, bad, stripes, monkeypatch):
    """k_frame_p5 (the packed-once pass as a streaming kernel: a wave per 62-item column, the five packed rows of the window in
    registers, pixel-map records collected per task, uncertain strips' tiles to the work list for the list-mode k_frame) takes long
    launches only; MLVFS_AMD_KF_P5=2 with MLVFS_AMD_KF_P=2 forces it (and the list-mode launch behind it) wherever it can run: one
    and several columns, a last column of one item, widths that are 8 mod 16, tasks of a few rows, every footage kind (adversarial:
    thousands of pixel-map cells and pixels at black -- the dense-map and the dark-row paths; colour patches: everything uncertain),
    pixel maps of both detection modes, with and without stripes."""
    from mlvfs_amd.stream import to_numpy_u16
    monkeypatch.setenv("MLVFS_AMD_KF_P", "2")
    monkeypatch.setenv("MLVFS_AMD_KF_P5", "2")
    for kind in ("normal", "low_light", "colour_cast", "adversarial"):
        if kind in ("low_light", "colour_cast"):
            frames = [getattr(synth, kind + "_frame")(w, h, seed=3 + k) for k in range(3)]
        else:
            frames = [getattr(synth, kind + "_frame")(w, h, frame=k) for k in range(3)]
        want, pixels, corr = oracle_clip(oracle, frames, w, h, 5, bad, stripes)
        s = make_stream(w, h)
        packed = s.upload_packed([synth.pack_bits(f) for f in frames])
        s.analyse_first_frame(packed, cs=5, bad_pix=bad, stripes=bool(stripes), rand_mode=1)
        got = to_numpy_u16(s.process(packed, cs=5, fix_pixels=bool(bad), stripes=bool(stripes)))
        s.close()
        for k in range(3):
            assert np.array_equal(got[k], want[k]), f"{kind} frame {k}: {(got[k] != want[k]).sum()} px differ"


@pytest.mark.parametrize("kind", ["normal", "low_light", "colour_cast"])
@pytest.mark.parametrize("cs,bad", [(5, 1), (5, 0), (2, 0)])
def test_long_launches_default_policy(torch_cuda, oracle, kind, cs, bad):
    """The library's own choice of kernels (no switches set) on launches long enough for the streaming kernels -- 3 600 frames of
    512x124: k_frame_p5 / k_frame_s take the first launch, the status words of finished launches then move low-light footage to
    k_frame_p + list and colour patches to k_frame alone (csrc/k_frame.hip: stream_state, stream_state_s).  Five launches in a row,
    the first three with the stream drained in between (every status word seen), the last two back to back: every launch's first,
    middle and last frames equal the oracle's."""
    import torch
    from mlvfs_amd.stream import to_numpy_u16
    for v in ("MLVFS_AMD_KF_P", "MLVFS_AMD_KF_P5", "MLVFS_AMD_KF_S"):
        assert v not in os.environ
    w, h, nf = 512, 124, 3600
    if kind == "normal":
        frames = [synth.normal_frame(w, h, frame=k) for k in range(8)]
    else:
        frames = [getattr(synth, kind + "_frame")(w, h, seed=5 + k) for k in range(8)]
    want, pixels, corr = oracle_clip(oracle, frames, w, h, cs, bad, 1)
    s = make_stream(w, h)
    base = s.upload_packed([synth.pack_bits(f) for f in frames])
    s.analyse_first_frame(base, cs=cs, bad_pix=bad, stripes=True, rand_mode=1)
    packed = s.alloc_packed(nf)
    for i in range(0, nf, 8):
        packed[i:i + 8] = base
    out = s.alloc_out(nf)
    sample = list(range(8)) + list(range(1796, 1804)) + list(range(nf - 8, nf))
    for launch in range(5):
        out.zero_()
        s.process(packed, out, cs=cs, fix_pixels=bool(bad), stripes=True)
        if launch < 3:
            torch.cuda.synchronize()
        got = to_numpy_u16(out[sample])
        for n, i in enumerate(sample):
            assert np.array_equal(got[n], want[i % 8]), f"{kind} cs{cs} launch {launch} frame {i}: {(got[n] != want[i % 8]).sum()} px differ"
    s.close()


def test_fused_adversarial(torch_cuda, oracle):
    """INT_MIN-heavy frame (40 % of pixels at black+-4) with thousands of bad pixels:
    exercises wrap-around EV arithmetic and multi-level ordered repair."""
    from mlvfs_amd.stream import to_numpy_u16
    w, h = 256, 130
    frames = [synth.adversarial_frame(w, h, frame=k) for k in range(2)]
    want, pixels, corr = oracle_clip(oracle, frames, w, h, 5, 2, 1)
    s = make_stream(w, h)
    packed = s.upload_packed([synth.pack_bits(f) for f in frames])
    s.analyse_first_frame(packed, cs=5, bad_pix=2, stripes=True)
    assert len(pixels) > 500 and np.array_equal(s.get_pixel_map(), pixels)
    got = to_numpy_u16(s.process(packed, cs=5, fix_pixels=True, stripes=True))
    for k in range(2):
        assert np.array_equal(got[k], want[k])
    s.close()


@pytest.mark.parametrize("cs", [0, 2, 3, 5])
@pytest.mark.parametrize("unpacked", [False, True])
def test_dense_focus_pixel_map(torch_cuda, oracle, cs, unpacked):
    """A focus-pixel map as some cameras have it: a regular grid, far more than 64 repaired cells per tile (the fused kernel
    takes the first 64 in its fourth wave during the loader phase, the rest afterwards), neighbours inside one cell and
    across tile borders, k_pixfix's flat grid over entries x frames."""
    from mlvfs_amd.stream import to_numpy_u16
    w, h = 416, 264
    frames = [synth.normal_frame(w, h, frame=k) for k in range(3)]
    ys, xs = np.mgrid[6:h - 6:3, 7:w - 8:5]
    pts = np.stack([xs.reshape(-1), ys.reshape(-1)], 1).astype(np.int32)
    edge = [[1, 50], [w - 2, 60], [100, 1], [120, h - 2], [2, 2], [w - 1, h - 1], [0, 100]]       # the edge rules of cs.c:479-497
    pts = np.concatenate([pts, pts[::7] + [1, 0], edge]).astype(np.int32)   # some cells with two repaired pixels
    assert len(pts) > 7000
    want = [oracle.apply_focus_pixels(f, BLACK, pts, (0, 0), 0) for f in frames]
    if cs:
        want = [oracle.chroma_smooth(f, BLACK, cs) for f in want]
    s = make_stream(w, h)
    s.set_pixel_map(pts, kind=1)
    packed = s.upload_packed([synth.pack_bits(f) for f in frames])
    if unpacked:
        got = to_numpy_u16(s.process_unpacked(s.unpack(packed), cs=cs, fix_pixels=True, stripes=False))
    else:
        got = to_numpy_u16(s.process(packed, cs=cs, fix_pixels=True, stripes=False))
    for k in range(3):
        assert np.array_equal(got[k], want[k]), f"frame {k}: {(got[k] != want[k]).sum()} px differ"
    # the in-place stage (scatter kernel) agrees
    staged = to_numpy_u16(s.fix_pixels(s.unpack(packed)))
    for k in range(3):
        assert np.array_equal(staged[k], oracle.apply_focus_pixels(frames[k], BLACK, pts, (0, 0), 0))
    s.close()


def test_stage_api_matches_fused(torch_cuda, oracle):
    """unpack_dev -> fix_pixels_dev -> chroma_smooth_dev -> stripes_apply_dev == fused launch."""
    from mlvfs_amd.stream import to_numpy_u16
    w, h = 416, 264
    frames = [synth.normal_frame(w, h, frame=k) for k in range(4)]
    s = make_stream(w, h)
    packed = s.upload_packed([synth.pack_bits(f) for f in frames])
    s.analyse_first_frame(packed, cs=5, bad_pix=1, stripes=True)
    staged = s.unpack(packed)
    assert np.array_equal(to_numpy_u16(staged), np.stack(frames))
    staged = s.stripes_apply(s.chroma_smooth(s.fix_pixels(staged), 5))
    fused = s.process(packed, cs=5, fix_pixels=True, stripes=True)
    assert np.array_equal(to_numpy_u16(staged), to_numpy_u16(fused))
    s.close()


def test_device_synth_equals_numpy(torch_cuda):
    """bench.py builds its stream on the GPU with the same generator the tests use on numpy."""
    w, h = 256, 128
    s = make_stream(w, h)
    dev = s.synth_packed(2, seed=1, first_frame=5).cpu().numpy()
    for k in range(2):
        words = synth.pack14(synth.normal_frame(w, h, seed=1, frame=5 + k)).astype("<u2").view(np.uint8)
        assert np.array_equal(dev[k, : words.size], words)
    s.close()


def test_row_sharded_histogram_is_shard_invariant(torch_cuda, oracle):
    """frame 0 row-sharded over G ranks -> per-shard accepted counts give the
    rand() offsets, partial histograms add up to the single-GPU histogram (here: shards run
    one after the other on one GPU; the integer adds commute)."""
    import torch
    from mlvfs_amd.stream import _dptr
    w, h = 256, 130
    f = synth.normal_frame(w, h)
    needed, coeffs, hist_ref, num_ref = oracle.stripes_compute(f, BLACK, WHITE, want_hist=True)
    s = make_stream(w, h)
    frame = torch.from_numpy(f.view(np.int16)).cuda()
    L = s.L
    for shards in (1, 2, 3):
        edges = [h * k // shards for k in range(shards + 1)]
        counts = []
        for k in range(shards):
            acc = C.c_int64(0)
            lib.check(L.mlvfs_amd_stripes_count_dev(C.byref(s.geom), _dptr(frame), edges[k], edges[k + 1], C.byref(acc), None))
            counts.append(acc.value)
        total = sum(counts)
        assert total == int(num_ref.sum())
        rnd = np.zeros(2 * total + 2, np.uint16)
        L.mlvfs_amd_rand_stream(lib.ptr(rnd), 2 * total, 0, 1)
        d_rnd = torch.from_numpy(rnd.view(np.int16)).cuda()
        d_hist = torch.zeros(8 * 65536, dtype=torch.int32, device="cuda")
        d_num = torch.zeros(8, dtype=torch.int32, device="cuda")
        off = 0
        for k in range(shards):
            acc = C.c_int64(0)
            lib.check(L.mlvfs_amd_stripes_count_dev(C.byref(s.geom), _dptr(frame), edges[k], edges[k + 1], C.byref(acc), None))
            lib.check(L.mlvfs_amd_stripes_hist_dev(C.byref(s.geom), _dptr(frame), edges[k], edges[k + 1],
                                                   C.c_void_p(d_rnd.data_ptr() + 4 * off), 2 * counts[k],
                                                   _dptr(d_hist), _dptr(d_num), None))
            off += counts[k]
        torch.cuda.synchronize()
        assert np.array_equal(d_hist.cpu().numpy().reshape(8, 65536), hist_ref)
        assert np.array_equal(d_num.cpu().numpy(), num_ref)
        co = np.zeros(8, np.int32)
        hh = np.ascontiguousarray(d_hist.cpu().numpy())
        nn = np.ascontiguousarray(d_num.cpu().numpy())
        assert L.mlvfs_amd_stripes_solve(lib.ptr(hh), lib.ptr(nn), w * h * 14 // 8, lib.ptr(co)) == needed
        assert list(co) == list(coeffs)
    s.close()


# ------------------------------------------------------------------ full size (BASELINE.json configs 2 and 3)
FULL_W, FULL_H = 3584, 1320


@pytest.mark.parametrize("cs", [2, 5])
def test_full_size_against_oracle(torch_cuda, oracle, cs):
    """One 3584x1320 frame through the fused path == oracle (cs5x5 takes ~1 s on the CPU)."""
    from mlvfs_amd.stream import to_numpy_u16
    w, h = FULL_W, FULL_H
    frames = [synth.normal_frame(w, h, frame=k) for k in range(2)]
    want, pixels, corr = oracle_clip(oracle, frames, w, h, cs, 1, 1)
    s = make_stream(w, h)
    packed = s.upload_packed([synth.pack14(f).astype("<u2") for f in frames])
    s.analyse_first_frame(packed, cs=cs, bad_pix=1, stripes=True)
    assert np.array_equal(s.get_pixel_map(), pixels)
    needed, co = s.get_stripes()
    assert needed == corr[0] == 1 and list(co) == list(corr[1])
    got = to_numpy_u16(s.process(packed, cs=cs, fix_pixels=True, stripes=True))
    for k in range(2):
        assert np.array_equal(got[k], want[k]), f"frame {k}: {(got[k] != want[k]).sum()} px differ"
    s.close()


def test_full_size_focus_pixel_map(torch_cuda, oracle):
    """3584x1320 with a focus-pixel map of the size some cameras have (every 8th column of every 12th row, 48 000 entries, plus
    neighbours three pixels apart that depend on each other), low-light footage (the spread table layout, shared references)."""
    from mlvfs_amd.stream import to_numpy_u16
    w, h = FULL_W, FULL_H
    frames = [synth.low_light_frame(w, h, seed=21 + k) for k in range(2)]
    ys, xs = np.mgrid[6:h - 6:12, 7:w - 8:8]
    pts = np.stack([xs.reshape(-1), ys.reshape(-1)], 1).astype(np.int32)
    pts = np.concatenate([pts, pts[::5] + [3, 0], pts[::9] + [0, 2]]).astype(np.int32)
    want = [oracle.chroma_smooth(oracle.apply_focus_pixels(f, BLACK, pts, (0, 0), 0), BLACK, 5) for f in frames]
    s = make_stream(w, h)
    s.set_pixel_map(pts, kind=1)
    packed = s.upload_packed([synth.pack14(f).astype("<u2") for f in frames])
    got = to_numpy_u16(s.process(packed, cs=5, fix_pixels=True, stripes=False))
    for k in range(2):
        assert np.array_equal(got[k], want[k]), f"frame {k}: {(got[k] != want[k]).sum()} px differ"
    assert s.get_t16_layout() == 2
    s.close()


def test_full_size_properties(torch_cuda):
    """Size-independent checks on a 16-frame 3584x1320 stream built in HBM:
      * unpack(pack(x)) == x for every frame (round trip),
      * batch invariance: frame k of a 16-frame launch == the same frame launched alone,
      * green pixels are never touched by chroma smoothing (SURVEY.md 8a notes 5),
      * stripes-apply with unit coefficients only clamps to the white level (stripes.c:263)."""
    import torch
    from mlvfs_amd.stream import to_numpy_u16
    w, h, n = FULL_W, FULL_H, 16
    s = make_stream(w, h)
    packed = s.synth_packed(n, seed=11)
    dev = torch.zeros(1, device="cuda")
    unp = s.unpack(packed)
    for k in (0, 7, 15):
        ref = synth.normal_frame(w, h, seed=11, frame=k, like=dev).to(torch.int16)
        assert torch.equal(unp[k].reshape(-1), ref.reshape(-1))
    s.set_stripes(1, [65536] * 8)
    batch = s.process(packed, cs=5, stripes=True)
    alone = s.process(packed[9:10], cs=5, stripes=True)
    assert torch.equal(batch[9], alone[0])
    plain = s.process(packed, cs=5)
    assert torch.equal(plain[:, 0::2, 1::2], unp[:, 0::2, 1::2])        # G1 untouched by chroma smoothing
    assert torch.equal(plain[:, 1::2, 0::2], unp[:, 1::2, 0::2])        # G2
    assert (plain != unp).any()
    assert torch.equal(batch, plain.clamp(max=WHITE))                   # unit gains: min(white, p)
    s.close()


@pytest.mark.parametrize("kind", ["normal", "adversarial", "low_light", "colour_cast"])
@pytest.mark.parametrize("cs", [2, 3, 5])
def test_t16_layouts_give_identical_results(torch_cuda, oracle, kind, cs):
    """The fused kernel keeps its raw2ev table in one of two LDS layouts (plain / spread for dark footage);
    either must reproduce the oracle, whatever the footage, and the automatic choice must pick the spread form for underexposed
    frames and the plain one for the benchmark's."""
    from mlvfs_amd.stream import to_numpy_u16
    w, h = 272, 136
    if kind == "low_light":
        frames = [synth.low_light_frame(w, h, seed=3 + k) for k in range(2)]
    elif kind == "colour_cast":
        frames = [synth.colour_cast_frame(w, h, seed=3 + k) for k in range(2)]
    else:
        frames = [getattr(synth, kind + "_frame")(w, h, seed=3, frame=k) for k in range(2)]
    want = [oracle.chroma_smooth(f.copy(), BLACK, cs) for f in frames]
    s = make_stream(w, h)
    packed = s.upload_packed([synth.pack_bits(f) for f in frames])
    for layout in (1, 2, 0):
        s.set_t16_layout(layout)
        got = to_numpy_u16(s.process(packed, cs=cs))
        for k in range(2):
            assert np.array_equal(got[k], want[k]), f"layout {layout} frame {k}: {(got[k] != want[k]).sum()} px differ"
    chosen = s.get_t16_layout()
    if kind == "low_light":
        assert chosen == 2
    if kind == "normal":
        assert chosen == 1
    # 16-bit input path
    unp = s.unpack(packed)
    for layout in (1, 2):
        s.set_t16_layout(layout)
        got = to_numpy_u16(s.process_unpacked(unp, cs=cs))
        for k in range(2):
            assert np.array_equal(got[k], want[k]), f"16-bit input, layout {layout} frame {k}"
    s.close()


@pytest.mark.parametrize("w,h", [(48, 20), (144, 40), (1296, 520)])
def test_tile_tickets_across_launch_sizes(torch_cuda, oracle, w, h):
    """k_frame hands out tiles by per-stream ticket counters that the last workgroup of a launch zeroes (DESIGN.md 3.1,
    Placement).  Launches of very different sizes back to back on one stream -- fewer tiles than workgroups, one tile per
    group, thousands of tiles -- must all process every tile exactly once: each launch is compared with the oracle and the
    output buffer is poisoned before it, so a tile that nobody drew (or a counter left over from the previous launch) shows."""
    import torch
    from mlvfs_amd.stream import to_numpy_u16
    nf = 37
    frames = [synth.normal_frame(w, h, seed=5, frame=k % 5) for k in range(nf)]
    want = [oracle.chroma_smooth(f.copy(), BLACK, 5) for f in frames[:5]]
    s = make_stream(w, h)
    packed = s.upload_packed([synth.pack_bits(f) for f in frames])
    out = s.alloc_out(nf)
    for lo, hi in [(0, 1), (0, 37), (3, 4), (1, 20), (36, 37), (0, 2), (5, 37), (0, 37)]:
        out.fill_(0x5A5A)
        s.process(packed[lo:hi], out[lo:hi], cs=5)
        got = to_numpy_u16(out[lo:hi])
        for k in range(lo, hi):
            assert np.array_equal(got[k - lo], want[k % 5]), f"launch [{lo},{hi}) frame {k}"
    s.close()


@pytest.mark.parametrize("pinned", [True, False])
def test_host_pipeline_matches_device_pipeline(gpu, pinned):
    """mlvfs_amd_process_frames_host (frames in host memory, chunked H2D / kernels / D2H on three streams) must give
    exactly what the device-resident pass gives, for chunk sizes that do and do not divide the frame count."""
    import torch
    from mlvfs_amd.stream import ClipStream
    w, h, n = 256, 130, 11
    s = ClipStream(w, h, 14, synth.BLACK, synth.WHITE, device=0)
    frames = [synth.normal_frame(w, h, frame=k) for k in range(n)]
    packed = s.upload_packed([synth.pack_bits(f) for f in frames])
    s.analyse_first_frame(packed, cs=5, bad_pix=1, stripes=True, rand_mode=1)
    as_bytes = lambda t: t.cpu().contiguous().view(torch.uint8).reshape(n, -1)
    want = as_bytes(s.process(packed, cs=5, fix_pixels=True, stripes=True))
    host_in = packed.cpu()
    if pinned:
        host_in = host_in.pin_memory()
    for chunk in (1, 4, 16):
        out = torch.zeros((n, s.out_stride), dtype=torch.uint8)
        if pinned:
            out = out.pin_memory()
        got = s.process_host(host_in, out, cs=5, fix_pixels=True, stripes=True, chunk=chunk)
        assert torch.equal(got, want)
    got = s.process_host(host_in, None, cs=0, fix_pixels=False, stripes=False, chunk=3)       # unpack only
    assert torch.equal(got, as_bytes(s.process(packed)))
    s.close()


@pytest.mark.gpu
@pytest.mark.parametrize("bpp", [12, 10])
def test_other_bit_depths_through_the_fused_entry_points(gpu, oracle, tmp_path, bpp):
    """10- and 12-bit clips (dng.c:813-843 unpacks any depth): device, host and file entry points equal the oracle's
    process_frame; the frames are unpacked to 16 bits first and then take the fused kernel's 16-bit input path."""
    from mlvfs_amd import mlvfile
    from mlvfs_amd.stream import ClipStream, to_numpy_u16
    w, h, n = 256, 130, 5
    shift = 14 - bpp
    black, white = synth.BLACK >> shift, synth.WHITE >> shift
    frames = [(synth.normal_frame(w, h, seed=4, frame=k) >> shift).astype(np.uint16) for k in range(n)]
    packed = [synth.pack_bits(f, bpp) for f in frames]
    s = ClipStream(w, h, bpp, black, white, device=0)
    dev_packed = s.upload_packed(packed)
    s.analyse_first_frame(dev_packed, cs=5, bad_pix=1, stripes=True, rand_mode=1)
    pixels = oracle.detect_bad_pixels(frames[0], black, 0)
    want, corr = [], None
    for f in frames:
        img = oracle.chroma_smooth(oracle.apply_bad_pixels(f, black, pixels), black, 5)
        if corr is None:
            corr = oracle.stripes_compute(img, black, white, frame_size=w * h * bpp // 8)
        want.append(oracle.stripes_apply(img, black, white, *corr))
    got = to_numpy_u16(s.process(dev_packed, cs=5, fix_pixels=True, stripes=True))
    for k in range(n):
        assert np.array_equal(got[k], want[k]), ("device", k)
    host_out = to_numpy_u16(s.process_host(dev_packed.cpu(), cs=5, fix_pixels=True, stripes=True, chunk=2))
    for k in range(n):
        assert np.array_equal(host_out[k].reshape(h, w), want[k]), ("host", k)
    names = mlvfile.write_clip(str(tmp_path / "B.MLV"), [p.tobytes() for p in packed], w, h, bpp=bpp, black=black, white=white)
    with mlvfile.MlvReader(names[0]) as r:
        out = np.zeros((n, h, w), np.uint16)
        r.process(s.clip, 0, n, out, cs=5, fix_pixels=True, stripes=True, batch=2)
        for k in range(n):
            assert np.array_equal(out[k], want[k]), ("file", k)
    s.close()


@pytest.mark.gpu
@pytest.mark.parametrize("bpp", [12, 10])
@pytest.mark.parametrize("w,h", [(256, 130), (264, 62), (128 + 16, 92), (400, 61), (1736, 64)])
def test_reduced_bit_depths_straight_into_the_fused_loader(gpu, oracle, bpp, w, h):
    """Round 4: 12-bit streams (8 pixels = 12 bytes) in rows of whole 8-pixel groups and 10-bit streams (8 pixels = 10 bytes) in rows
    of whole 16-pixel groups are read by the fused kernel's loader itself (k_frame.hip, VEC = 3 / 4); other geometries still take the
    unpack pass first.  Either way: every method, with the pixel map, with pixels at and below black, equals the oracle."""
    from mlvfs_amd.stream import ClipStream, to_numpy_u16
    shift = 14 - bpp
    black, white = synth.BLACK >> shift, synth.WHITE >> shift
    frames = []
    for k in range(3):
        f = (synth.normal_frame(w, h, seed=11, frame=k, hot=30, cold=30) >> shift).astype(np.uint16)
        rng = np.random.default_rng(100 * bpp + k)
        ys, xs = rng.integers(0, h, 40), rng.integers(0, w, 40)
        f[ys[:20], xs[:20]] = black                                   # ev = INT_MIN
        f[ys[20:], xs[20:]] = max(black - 3, 0)                       # below black
        frames.append(f)
    s = ClipStream(w, h, bpp, black, white, device=0)
    dev_packed = s.upload_packed([synth.pack_bits(f, bpp) for f in frames])
    pixels = oracle.detect_bad_pixels(frames[0], black, 0)
    s.set_pixel_map(pixels)
    co = [65536, 65536, 65354, 65738, 65241, 65868, 65450, 65640]
    s.set_stripes(1, co)
    for cs in (0, 2, 3, 5):
        got = to_numpy_u16(s.process(dev_packed, cs=cs, fix_pixels=True, stripes=True))
        for k, f in enumerate(frames):
            img = oracle.apply_bad_pixels(f, black, pixels)
            if cs:
                img = oracle.chroma_smooth(img, black, cs)
            img = oracle.stripes_apply(img, black, white, 1, np.array(co, np.int32)) if w % 8 == 0 else img
            assert np.array_equal(got[k], img), (bpp, w, h, cs, k, int((got[k] != img).sum()))
    s.close()


@pytest.mark.gpu
def test_deflicker_equals_reference_histogram_and_formula(gpu, reference):
    """main.c:895-906 (static in main.c, restated here) on the reference's own histogram helpers (oracle/_ref): every second
    pixel from pixel 1, 16-bit counters that wrap, median, BaselineExposure numerator."""
    import torch
    w, h = 3584, 1320                                     # large enough for the 16-bit counters to wrap
    for kind, seed in (("normal", 3), ("adversarial", 5)):
        f = getattr(synth, kind + "_frame")(w, h, seed=seed)
        flat = np.ascontiguousarray(f.reshape(-1))
        size_bytes = flat.size * 2
        n = (size_bytes - 1) // 2
        white = (1 << 14) + 1
        median = reference.L.ref_hist_median_of(np.ascontiguousarray(flat[1:]), n, 1, white)
        d = torch.from_numpy(flat.view(np.int16)).cuda()
        geom = lib.Geom(w, h, 14, synth.BLACK, synth.WHITE, 0, 0)
        for target in (3072, 5000, synth.BLACK):
            want = np.float64(target - synth.BLACK) / np.float64(int(median) - synth.BLACK)
            with np.errstate(divide="ignore"):
                corr = np.log2(want) * 10000
            want0 = int(np.trunc(corr)) if np.isfinite(corr) else -2 ** 31
            eb = np.zeros(2, np.int32)
            lib.check(gpu.mlvfs_amd_deflicker_dev(C.byref(geom), C.c_void_p(d.data_ptr()), size_bytes, target, lib.ptr(eb), None))
            assert (int(eb[0]), int(eb[1])) == (want0, 10000), (kind, target, median)


@pytest.mark.gpu
def test_device_rand_stream_equals_host_stream(gpu):
    """The rand() % 1024 dither stream generated on the device (jump-ahead per 992-value chunk) is the host stream, which
    tests/test_cabi.py pins against libc's rand()."""
    import torch
    for n, skip in ((1, 0), (7, 0), (992, 0), (993, 5), (100_003, 0), (5_000_000, 12_345_678), (28_000_010, 0), (4096, 2 ** 33 + 17)):
        want = np.zeros(n, np.uint16)
        gpu.mlvfs_amd_rand_stream(lib.ptr(want), n, skip, 1)
        d = torch.full((n + 16,), -1, dtype=torch.int16, device="cuda")
        lib.check(gpu.mlvfs_amd_rand_stream_dev(C.c_void_p(d.data_ptr()), n, skip, 1, None))
        got = d.cpu().numpy().view(np.uint16)
        assert np.array_equal(got[:n], want), (n, skip)
        assert (got[n:] == 0xFFFF).all(), "wrote past n"

"""LJ92 payload decoder.

CPU: the restatement oracle/oracle_lj92.c against the reference's own lj92.c (oracle/_ref) on streams from the reference's
encoder and on hand-made streams with every predictor; the host parser of the library.
GPU: csrc/k_lj92.hip against the oracle (and, where present, the reference), bit for bit."""
import numpy as np
import pytest

from mlvfs_amd import synth
from oracle import lj92_testenc as enc


def quadrants(f):
    """What an MLV writer compresses: the four Bayer channels as the quadrants of one image."""
    return np.ascontiguousarray(np.block([[f[0::2, 0::2], f[0::2, 1::2]], [f[1::2, 0::2], f[1::2, 1::2]]]))


def images(w, h, seed=3):
    rng = np.random.default_rng(seed)
    f = synth.normal_frame(w, h, seed=seed)
    return {"smooth": quadrants(f), "noise": rng.integers(0, 16384, (h, w)).astype(np.uint16), "flat": np.full((h, w), 2048, np.uint16),
            "dark": quadrants(synth.adversarial_frame(w, h, seed=seed))}


SIZES = [(64, 48), (136, 72), (256, 130)]


@pytest.mark.parametrize("w,h", SIZES)
def test_oracle_equals_reference(oracle, reference, w, h):
    for name, img in images(w, h).items():
        s = reference.lj92_encode(img, 14)                                     # lj92.c:1104-1146: predictor 6
        st_r, dr = reference.lj92_decode(s)
        st_o, do = oracle.lj92_decode(s)
        assert st_r == st_o == 0 and np.array_equal(dr, img) and np.array_equal(do, img), name
        assert oracle.lj92_info(s) == dict(width=w, height=h, bits=14, predictor=6, huffbits=oracle.lj92_info(s)["huffbits"],
                                           scan_offset=oracle.lj92_info(s)["scan_offset"])
        for p in range(8):                                                     # every branch of parseScan (lj92.c:546-563)
            s = enc.encode(img, p, 14, comment=b"made by tests" if p == 3 else None)
            st_r, dr = reference.lj92_decode(s)
            st_o, do = oracle.lj92_decode(s)
            assert st_r == st_o == 0 and np.array_equal(dr, do), (name, p)
            assert np.array_equal(do, img), (name, p)
    wide = np.random.default_rng(1).integers(0, 65536, (h, w)).astype(np.uint16)
    for p in (1, 2, 3):                                                        # 16-bit samples, differences up to 16 bits long
        s = enc.encode(wide, p, 16)
        st_r, dr = reference.lj92_decode(s)
        st_o, do = oracle.lj92_decode(s)
        assert st_r == st_o == 0 and np.array_equal(dr, do) and np.array_equal(do, wide), p


GOLD = None


def gold():
    global GOLD
    if GOLD is None:
        import os
        GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lj92_vectors.npz"))
    return GOLD


def test_oracle_reproduces_reference_vectors(oracle):
    """tests/golden/lj92_vectors.npz: streams and the images the reference's decoder made of them."""
    g = gold()
    keys = sorted(k[:-7] for k in g.files if k.endswith("_stream"))
    assert len(keys) >= 40
    for k in keys:
        st, img = oracle.lj92_decode(g[k + "_stream"].tobytes())
        assert st == 0 and np.array_equal(img, g[k + "_image"]), k


@pytest.mark.gpu
def test_gpu_reproduces_reference_vectors(gpu, oracle):
    g = gold()
    keys = sorted(k[:-7] for k in g.files if k.endswith("_stream"))
    assert len(keys) >= 40
    for k in keys:
        img = g[k + "_image"]
        h, w = img.shape
        got = gpu_decode([g[k + "_stream"].tobytes()], w, h)[0]
        assert np.array_equal(got, oracle.lj92_untile(img, w, h)), k


def test_untile_restatement(oracle):
    w, h = 64, 48
    f = synth.normal_frame(w, h, seed=2)
    assert np.array_equal(oracle.lj92_untile(quadrants(f), w, h), f)           # main.c:646-667 undoes the quadrant layout
    # the JPEG's own dimensions do not matter, only the flat order does
    assert np.array_equal(oracle.lj92_untile(quadrants(f).reshape(h // 2, w * 2), w, h), f)


def test_host_parser(amd, reference):
    from mlvfs_amd import lj92
    img = images(64, 48)["smooth"]
    assert lj92.info(reference.lj92_encode(img, 14)) == dict(width=64, height=48, bits=14, predictor=6)
    assert lj92.info(enc.encode(img, 1, 12, comment=b"x" * 40)) == dict(width=64, height=48, bits=12, predictor=1)
    for junk in (b"", b"\xff\xd8\xff\xd9", b"not a jpeg at all, not even close", enc.encode(img, 6, 14)[:30]):
        with pytest.raises(Exception):
            lj92.info(junk)


def test_dropin_lj92_open_on_the_cpu(amd, reference):
    """lj92_open / lj92_close of the library need no GPU: dimensions and bit depth as the reference's lj92_open reports them
    (main.c:626), damage refused, a null handle tolerated by lj92_close like free()."""
    import ctypes as C
    for w, h, bits, shape in ((64, 48, 14, (48, 64)), (256, 130, 14, (65, 512)), (136, 72, 12, (72, 136))):
        img = (images(w, h)["smooth"] >> (14 - bits)).astype(np.uint16)
        s = reference.lj92_encode(np.ascontiguousarray(img.reshape(shape)), bits)
        buf = np.frombuffer(s, np.uint8).copy()
        hd = C.c_void_p()
        a, b, c = C.c_int(), C.c_int(), C.c_int()
        assert amd.lj92_open(C.byref(hd), C.c_void_p(buf.ctypes.data), buf.size, C.byref(a), C.byref(b), C.byref(c)) == 0 and hd.value
        rst, rimg = reference.lj92_decode(s)
        assert rst == 0 and (a.value, b.value, c.value) == (rimg.shape[1], rimg.shape[0], bits)
        amd.lj92_close(hd)
    for junk in (b"\xff\xd8\xff\xd9", b"not a jpeg at all, not even close"):
        buf = np.frombuffer(junk, np.uint8).copy()
        hd = C.c_void_p(1)
        assert amd.lj92_open(C.byref(hd), C.c_void_p(buf.ctypes.data), buf.size, None, None, None) != 0 and not hd.value
    amd.lj92_close(None)


# ---------------------------------------------------------------- GPU
def gpu_decode(streams, xres, yres):
    from mlvfs_amd import lj92
    return lj92.decode_frames(streams, xres, yres).cpu().numpy().view(np.uint16)


def want(oracle, stream, xres, yres):
    st, img = oracle.lj92_decode(stream)
    assert st == 0
    return oracle.lj92_untile(img, xres, yres)


@pytest.mark.gpu
@pytest.mark.parametrize("w,h", SIZES + [(18, 6), (2, 2), (1920, 18)])
def test_gpu_equals_oracle(gpu, oracle, w, h):
    rng = np.random.default_rng(w * h)
    imgs = images(w, h) if w >= 64 else {"noise": rng.integers(0, 16384, (h, w)).astype(np.uint16)}
    streams, names = [], []
    for name, img in imgs.items():
        for p in ((6, 1) if name != "noise" else range(8)):                    # every predictor of lj92.c:546-563 on the noise image
            streams.append(enc.encode(img, p, 14, comment=b"c" if p == 1 else None))
            names.append((name, p))
    streams.append(enc.encode(imgs["noise"], 6, 14, ramp=True))               # codes up to 15 bits: the table stays in global memory
    names.append(("noise-ramp", 6))
    got = gpu_decode(streams, w, h)                                            # one batch, streams of different lengths
    for k, s in enumerate(streams):
        assert np.array_equal(got[k], want(oracle, s, w, h)), names[k]


@pytest.mark.gpu
def test_gpu_reference_encoder_streams_and_odd_shapes(gpu, oracle, reference):
    w, h = 256, 130
    f = synth.normal_frame(w, h, seed=9)
    q = quadrants(f)
    for shape in ((h, w), (h // 2, w * 2), (h * 2, w // 2)):                   # JPEG dimensions != video dimensions (main.c:646-667)
        s = reference.lj92_encode(np.ascontiguousarray(q.reshape(shape)), 14)
        got = gpu_decode([s], w, h)[0]
        assert np.array_equal(got, f) and np.array_equal(got, want(oracle, s, w, h)), shape
    wide = np.random.default_rng(5).integers(0, 65536, (h, w)).astype(np.uint16)   # 0xFF bytes galore, 16-bit differences
    for p in (1,):
        s = enc.encode(wide, p, 16)
        assert s.count(b"\xff\x00") > 50
        assert np.array_equal(gpu_decode([s], w, h)[0], want(oracle, s, w, h))


def dropin_decode(gpu, stream):
    """The reference decoder's own three calls (lj92.h:40-58) as the library exports them, the way main.c:626-647 uses them."""
    import ctypes as C
    buf = np.frombuffer(stream, np.uint8).copy()
    h = C.c_void_p()
    w_, h_, b_ = C.c_int(), C.c_int(), C.c_int()
    st = gpu.lj92_open(C.byref(h), C.c_void_p(buf.ctypes.data), buf.size, C.byref(w_), C.byref(h_), C.byref(b_))
    if st != 0:
        return st, None, None
    out = np.zeros((h_.value, w_.value), np.uint16)
    st = gpu.lj92_decode(h, C.c_void_p(out.ctypes.data), out.size, 0, None, 0)
    gpu.lj92_close(h)
    return st, out, (w_.value, h_.value, b_.value)


@pytest.mark.gpu
def test_decode_untiled_equals_decode_plus_the_callers_loop(gpu, oracle):
    """mlvfs_amd_lj92_decode_untiled (optional: three changed lines in main.c) = lj92_decode + the untiling loop of main.c:646-667."""
    import ctypes as C
    for (w, h, seed) in ((256, 130, 9), (64, 48, 3), (1920, 1080, 5)):
        f = synth.normal_frame(w, h, seed=seed)
        q = quadrants(f)
        for shape in ((h, w), (h // 2, w * 2)):
            s = enc.encode(np.ascontiguousarray(q.reshape(shape)), 6, 14)
            buf = np.frombuffer(s, np.uint8).copy()
            hd = C.c_void_p()
            w_, h_, b_ = C.c_int(), C.c_int(), C.c_int()
            assert gpu.lj92_open(C.byref(hd), C.c_void_p(buf.ctypes.data), buf.size, C.byref(w_), C.byref(h_), C.byref(b_)) == 0
            out = np.zeros((h, w), np.uint16)
            gpu.mlvfs_amd_lj92_decode_untiled.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
            assert gpu.mlvfs_amd_lj92_decode_untiled(hd, C.c_void_p(out.ctypes.data), w, h) == 0
            assert gpu.mlvfs_amd_lj92_decode_untiled(hd, C.c_void_p(out.ctypes.data), w + 2, h) == -1      # another geometry than the JPEG's
            gpu.lj92_close(hd)
            assert np.array_equal(out, f), (w, h, shape)


@pytest.mark.gpu
def test_dropin_lj92_symbols_equal_the_reference_decoder(gpu, oracle, reference):
    """lj92_open / lj92_decode / lj92_close of the library against the reference's (oracle/_ref builds lj92.c): dimensions, every
    value in the decoder's own order (main.c untiles afterwards), the reference encoder's streams in three JPEG shapes, every
    predictor, a full-size frame; damage and the arguments MLVFS never passes come back as errors."""
    import ctypes as C
    w, h = 256, 130
    f = synth.normal_frame(w, h, seed=9)
    q = quadrants(f)
    for shape in ((h, w), (h // 2, w * 2), (h * 2, w // 2)):
        s = reference.lj92_encode(np.ascontiguousarray(q.reshape(shape)), 14)
        st, got, dims = dropin_decode(gpu, s)
        rst, rimg = reference.lj92_decode(s)
        assert st == 0 and rst == 0 and dims[:2] == (rimg.shape[1], rimg.shape[0]) and dims[2] == 14 and np.array_equal(got, rimg), shape
        assert np.array_equal(oracle.lj92_untile(got, w, h), f)                                   # what main.c's loop makes of it
    noise = images(136, 72)["noise"]
    for p in range(8):
        s = enc.encode(noise, p, 14)
        st, got, dims = dropin_decode(gpu, s)
        ost, oimg = oracle.lj92_decode(s)
        assert st == 0 and ost == 0 and np.array_equal(got, oimg), p
    big = synth.normal_frame(3584, 1320, seed=1)
    s = reference.lj92_encode(quadrants(big), 14)
    st, got, dims = dropin_decode(gpu, s)
    assert st == 0 and np.array_equal(oracle.lj92_untile(got, 3584, 1320), big)
    # errors
    good = enc.encode(images(136, 72)["smooth"], 6, 14)
    assert dropin_decode(gpu, good[: len(good) // 2])[0] != 0                 # the data ends before the last pixel
    assert dropin_decode(gpu, b"\xff\xd8 no jpeg")[0] != 0
    buf = np.frombuffer(good, np.uint8).copy()
    hd = C.c_void_p()
    assert gpu.lj92_open(C.byref(hd), C.c_void_p(buf.ctypes.data), buf.size, None, None, None) == 0
    out = np.zeros(136 * 72, np.uint16)
    lin = np.zeros(1 << 14, np.uint16)          # (the first row of a predictor-6 stream goes through the table unchecked, lj92.c:436-441)
    assert gpu.lj92_decode(hd, C.c_void_p(out.ctypes.data), out.size, 0, C.c_void_p(lin.ctypes.data), 16) != 0     # a value beyond the table: corrupt, like the reference
    assert gpu.lj92_decode(hd, C.c_void_p(out.ctypes.data), out.size - 1, 0, None, 0) != 0         # target too small
    assert gpu.lj92_decode(hd, C.c_void_p(out.ctypes.data), out.size, 0, None, 0) == 0
    gpu.lj92_close(hd)
    assert gpu.lj92_decode(None, C.c_void_p(out.ctypes.data), out.size, 0, None, 0) != 0


@pytest.mark.gpu
@pytest.mark.parametrize("pred", [1, 4, 6, 7])
def test_dropin_lj92_decode_with_skip_length_and_linearisation_table(gpu, reference, pred):
    """The two arguments MLVFS never passes (main.c:626-647), against the reference's own lj92_decode (lj92.c:436-493, 517-585): blocks of
    `writeLength` values `skipLength` apart (a tile written into a wider image), every value through a table; a block length of one;
    a value beyond the table.  VERDICT r4 missing #4."""
    import ctypes as C
    w, h = 136, 72
    img = images(w, h)["noise"]
    s = enc.encode(img, pred, 14)
    buf = np.frombuffer(s, np.uint8).copy()
    RL = reference.L
    for L in (RL, gpu):
        L.lj92_open.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.lj92_decode.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int]
        L.lj92_close.argtypes = [C.c_void_p]
    rng = np.random.default_rng(5)
    table = rng.integers(0, 65535, 1 << 14).astype(np.uint16)
    short = table.copy()                            # (passed with a length the frame's values go beyond)
    cases = [(w, 24, None, 0), (w // 2, 7, None, 0), (1, 3, None, 0), (w * h, 0, table, table.size), (w, 16, table, table.size), (w, 0, short, 9000)]
    for wl, sl, lin, linlen in cases:
        nblocks = -(-w * h // wl)
        size = w * h + nblocks * sl + 64
        res = []
        for L in (RL, gpu):
            hd = C.c_void_p()
            ww, hh, bb = C.c_int(), C.c_int(), C.c_int()
            assert L.lj92_open(C.byref(hd), C.c_void_p(buf.ctypes.data), buf.size, C.byref(ww), C.byref(hh), C.byref(bb)) == 0
            out = np.full(size, 0xABCD, np.uint16)
            st = L.lj92_decode(hd, C.c_void_p(out.ctypes.data), wl, sl, None if lin is None else C.c_void_p(lin.ctypes.data), linlen)
            L.lj92_close(hd)
            res.append((st != 0, out))
        assert res[0][0] == res[1][0], (wl, sl, linlen)
        if not res[0][0]:
            assert np.array_equal(res[0][1], res[1][1]), (wl, sl, linlen)
        else:                                       # (the reference stops where the value leaves the table: what it wrote until then is the same)
            n = int(np.argmax(res[0][1] == 0xABCD)) if (res[0][1] == 0xABCD).any() else size
            assert np.array_equal(res[0][1][:n], res[1][1][:n])


@pytest.mark.gpu
def test_gpu_full_size_frame(gpu, oracle, reference):
    w, h = 3584, 1320
    frames = [synth.normal_frame(w, h, seed=1, frame=k) for k in range(2)]
    streams = [reference.lj92_encode(quadrants(f), 14) for f in frames]
    got = gpu_decode(streams, w, h)
    for k in range(2):
        assert np.array_equal(got[k], frames[k])
    assert np.array_equal(got[0], want(oracle, streams[0], w, h))


@pytest.mark.gpu
def test_gpu_workgroups_with_more_symbols_than_their_stage_holds(gpu, oracle, reference):
    """k_lj_decode collects a workgroup's differences (8 KiB of stream) in LDS, 9 184 of them; highly compressible content -- flat
    areas at 4 bits per symbol with the reference's own table, 16 000 symbols per workgroup -- overflows that stage and the rest takes the direct path.  Frames
    that mix flat halves, noise (8 000 symbols per workgroup) and a ramp, in one batch with different lengths."""
    w, h = 1536, 640
    rng = np.random.default_rng(5)
    yy, xx = np.mgrid[0:h, 0:w]
    flat = np.full((h, w), 3000, np.uint16)
    half = flat.copy(); half[:, w // 2:] = rng.integers(0, 16384, (h, w - w // 2))
    rows = flat.copy(); rows[h // 3: 2 * h // 3] = rng.integers(2000, 2400, (2 * h // 3 - h // 3, w))
    ramp = (2048 + 3 * xx + yy).clip(0, 16383).astype(np.uint16)
    frames = [flat, half, rows.astype(np.uint16), ramp]
    streams = [reference.lj92_encode(quadrants(f), 14) for f in frames]
    assert len(streams[0]) * 8 / (w * h) < 4.5                               # (the flat frame: 16 000 symbols per 8 KiB, the stage holds 9 184)
    got = gpu_decode(streams, w, h)
    for k, f in enumerate(frames):
        assert np.array_equal(got[k], f), k
    assert np.array_equal(got[1], want(oracle, streams[1], w, h))


@pytest.mark.gpu
def test_gpu_rejects_damage_and_unsupported(gpu, oracle):
    w, h = 136, 72
    img = images(w, h)["smooth"]
    good = enc.encode(img, 6, 14)
    assert np.array_equal(gpu_decode([good], w, h)[0], want(oracle, good, w, h))
    with pytest.raises(Exception, match="damaged"):
        gpu_decode([good[: len(good) // 2]], w, h)                             # the data ends before the last pixel
    for p in (0, 2, 3, 4, 5, 7):                                               # the rarely used predictors, smooth image
        s = enc.encode(img, p, 14)
        assert np.array_equal(gpu_decode([s], w, h)[0], want(oracle, s, w, h)), p
    with pytest.raises(Exception, match="video frame"):
        gpu_decode([good], w + 2, h)


def test_host_parser_survives_mutations(amd):
    """Random damage to a valid stream's first bytes: the host parser answers or refuses, it never reads out of bounds
    (run under the normal allocator: a crash here is the failure)."""
    from mlvfs_amd import lj92
    rng = np.random.default_rng(7)
    good = np.frombuffer(enc.encode(images(64, 48)["smooth"], 6, 14, comment=b"x"), np.uint8)
    answered = 0
    for _ in range(3000):
        s = good.copy()
        for _ in range(int(rng.integers(1, 6))):
            s[int(rng.integers(0, 120))] = int(rng.integers(0, 256))
        cut = int(rng.integers(0, 4))
        s = s[: len(s) - [0, 1, len(s) - 40, len(s) - 70][cut]] if cut else s
        try:
            d = lj92.info(s.tobytes())
            answered += 1
            assert 0 < d["width"] <= 65535 and 0 < d["height"] <= 65535 and 0 <= d["predictor"] <= 7
        except lib_error():
            pass
    assert answered > 100


def lib_error():
    from mlvfs_amd import lib
    return lib.MlvfsAmdError


@pytest.mark.gpu
def test_gpu_survives_damaged_streams(gpu, oracle):
    """Damaged entropy-coded data and Huffman tables: the call fails or returns pixels, the GPU neither hangs nor faults,
    and an undamaged stream decodes correctly afterwards."""
    from mlvfs_amd import lib
    w, h = 136, 72
    img = images(w, h)["smooth"]
    good = np.frombuffer(enc.encode(img, 6, 14), np.uint8)
    rng = np.random.default_rng(11)
    failed = 0
    for trial in range(60):
        s = good.copy()
        lo = 20 if trial % 2 else 70                                           # with / without hits in the table and frame header
        for _ in range(int(rng.integers(1, 20))):
            s[int(rng.integers(lo, len(s)))] = int(rng.integers(0, 256))
        try:
            gpu_decode([s.tobytes()], w, h)
        except lib.MlvfsAmdError:
            failed += 1
    assert failed > 0
    assert np.array_equal(gpu_decode([good.tobytes()], w, h)[0], want(oracle, good.tobytes(), w, h))

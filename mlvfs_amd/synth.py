"""Deterministic synthetic MLV frame payloads (the repo's own generator).

There is no network and the reference ships no sample MLV, so every test and the
benchmark run on seeded synthetic 14-bit Bayer frames.  The generator is
counter-based (a 32-bit integer hash of seed, frame number and pixel index), so
the same code runs vectorised on numpy arrays (tests, fixtures) and on torch
tensors resident in HBM (bench.py builds its 1000-frame stream on the GPU).

Frame kinds
-----------
normal       smooth RGGB gradient, per-column-phase gains (period 8, makes the
             vertical-stripe correction fire), +-32 noise, K hot and K cold
             pixels incl. pairs two pixels apart (exercises the ordered
             bad-pixel repair, mlvfs/cs.c:314-330).
adversarial  40 % of pixels within black+-4 (raw2ev[black] is INT_MIN in the
             reference, mlvfs/main.c:163-167), 1 % below black, 2 % >= 16380.
dual_iso     two dark / two bright rows interleaved (3 EV apart) with clipping.

Packing follows mlvfs/raw.h:41-79: pixels form an MSB-first bit stream stored
as little-endian 16-bit words; the reference reads two pixels past the end of
a frame (mlvfs/main.c:579), hence the zero pad.
"""
from __future__ import annotations

import numpy as np

BLACK = 2048
WHITE = 15000
GAINS = (1.0, 1.0, 1.01, 0.99, 1.015, 0.985, 1.005, 0.995)
_M32 = 0xFFFFFFFF


def _is_torch(x) -> bool:
    return type(x).__module__.startswith("torch")


def _mix32(x):
    """32-bit integer hash on int64 containers (products stay below 2**63)."""
    x = ((x ^ (x >> 16)) * 0x45D9F3B) & _M32
    x = ((x ^ (x >> 16)) * 0x45D9F3B) & _M32
    return x ^ (x >> 16)


def _index_grid(w: int, h: int, like=None):
    if like is not None and _is_torch(like):
        import torch
        idx = torch.arange(w * h, dtype=torch.int64, device=like.device)
    else:
        idx = np.arange(w * h, dtype=np.int64)
    return idx, idx % w, idx // w


def _rand32(idx, seed: int, frame: int, salt: int):
    key = (seed * 0x9E3779B1 + frame * 0x85EBCA6B + salt * 0xC2B2AE35) & _M32
    return _mix32((idx + key) & _M32)


def _clip(x, lo, hi):
    if _is_torch(x):
        return x.clamp(lo, hi)
    return np.clip(x, lo, hi)


def _where(c, a, b):
    if _is_torch(c):
        import torch
        if not _is_torch(a):
            a = torch.full_like(b, a)
        return torch.where(c, a, b)
    return np.where(c, a, b)


def normal_frame(w: int, h: int, seed: int = 1, frame: int = 0, hot: int = 64, cold: int = 64,
                 black: int = BLACK, like=None):
    """16-bit container frame (values 0..16383), RGGB, see module docstring."""
    idx, x, y = _index_grid(w, h, like)
    # fixed-point gains (x1000) keep the arithmetic integer and identical on
    # numpy and torch
    phase = x & 7
    gain = 0
    for k, g in enumerate(GAINS):
        gk = int(round(g * 1000))
        gain = gain + (phase == k) * gk
    base = 200 + (6000 * x) // w + (2000 * y) // h          # linear value above black
    v = (base * gain) // 1000 + black + (_rand32(idx, seed, frame, 1) % 64) - 32
    v = _clip(v, 0, 16383)
    # defects: positions drawn from the hash, kept >= 8 px from the borders; every
    # 4th defect gets a twin two pixels to the right (same colour plane)
    n_def = hot + cold
    if n_def and w > 16 and h > 16:          # frames too small to keep defects 8 px from the borders get none
        k = np.arange(n_def, dtype=np.int64)
        r = _mix32((k * 2654435761 + seed * 97 + frame * 7919 + 12345) & _M32)
        dx = 8 + (r % (w - 16))
        dy = 8 + ((r >> 11) % (h - 16))
        vals = np.where(k < hot, 16000, black - 200)
        twin = (k % 4) == 0
        pos = np.concatenate([dx + dy * w, (np.minimum(dx + 2, w - 9) + dy * w)[twin]])
        val = np.concatenate([vals, vals[twin]])
        if _is_torch(v):
            import torch
            v[torch.as_tensor(pos, device=v.device)] = torch.as_tensor(val, device=v.device)
        else:
            v[pos] = val
    return _to_u16(v).reshape(h, w)


def adversarial_frame(w: int, h: int, seed: int = 7, frame: int = 0, black: int = BLACK, like=None):
    idx, x, y = _index_grid(w, h, like)
    r = _rand32(idx, seed, frame, 2)
    sel = r % 100
    body = 200 + (9000 * ((x * 7 + y * 3) % 512)) // 512 + black + ((r >> 8) % 128) - 64
    near_black = black - 4 + ((r >> 9) % 9)                  # black-4 .. black+4
    below = black - 5 - ((r >> 9) % 300)
    top = 16380 + ((r >> 9) % 4)
    v = _where(sel < 40, near_black, body)
    v = _where(sel == 40, below, v)
    v = _where((sel == 41) | (sel == 42), top, v)
    return _to_u16(_clip(v, 0, 16383)).reshape(h, w)


def dual_iso_frame(w: int, h: int, seed: int = 3, frame: int = 0, black: int = BLACK, like=None):
    """Rows with y%4 in {2,3} are the bright exposure (x8), RGGB gains .6/1/1/.5."""
    idx, x, y = _index_grid(w, h, like)
    r = _rand32(idx, seed, frame, 3)
    if _is_torch(x):
        import torch
        xf, yf = x.to(torch.float64), y.to(torch.float64)
        scene = 30 + 1500 * (0.5 + 0.5 * torch.sin(0.013 * xf) * torch.cos(0.017 * yf))
        scene = scene.to(torch.int64)
    else:
        scene = (30 + 1500 * (0.5 + 0.5 * np.sin(0.013 * x) * np.cos(0.017 * y))).astype(np.int64)
    scene = scene + 600 * (((x // 64) + (y // 64)) & 1)
    cfa = (y & 1) * 2 + (x & 1)                              # 0=R 1=G1 2=G2 3=B
    gain10 = (cfa == 0) * 6 + (cfa == 1) * 10 + (cfa == 2) * 10 + (cfa == 3) * 5
    lin = (scene * gain10) // 10
    bright = (y % 4) >= 2
    lin = _where(bright, lin * 8, lin) + (r % 33) - 16
    clip_at = 15200 - black + ((r >> 8) % 16)
    lin = _where(lin > clip_at, clip_at, lin)
    return _to_u16(_clip(lin + black, 0, 16383)).reshape(h, w)


def _to_u16(v):
    if _is_torch(v):
        import torch
        # torch has no uint16 arithmetic: device frames are int32 containers (0..16383)
        return v.to(torch.int32)
    return v.astype(np.uint16)


def packed_words(npix: int, bpp: int = 14) -> int:
    """16-bit words the reference reads for npix pixels (+2 px pad, main.c:579)."""
    return (npix + 2) * bpp // 16


def pack_bits(frame, bpp: int = 14) -> np.ndarray:
    """numpy: pack pixels MSB-first into LE 16-bit words, padded as the reference reads."""
    px = np.asarray(frame).astype(np.uint32).ravel()
    n = px.size
    shifts = np.arange(bpp - 1, -1, -1, dtype=np.uint32)
    bits = ((px[:, None] >> shifts) & 1).astype(np.uint8).ravel()
    total_words = packed_words(n, bpp) + 1
    pad = total_words * 16 - bits.size
    bits = np.concatenate([bits, np.zeros(pad, np.uint8)])
    be = np.packbits(bits).view(">u2")
    return be.astype("<u2")[: total_words]


def pack14(frame):
    """14-bit packing with plain integer ops; works on numpy arrays and torch tensors.

    Returns one frame's 16-bit words (npix*14/16 of them, no pad) as int32 values
    in 0..65535; npix must be a multiple of 8.
    """
    if _is_torch(frame):
        import torch
        p = frame.reshape(-1, 8).to(torch.int64)
        stack = torch.stack
    else:
        p = np.asarray(frame).astype(np.int64).reshape(-1, 8)
        stack = np.stack
    a, b, c, d, e, f, g, hh = (p[:, i] for i in range(8))
    words = stack([
        (a << 2) | (b >> 12),
        ((b & 0xFFF) << 4) | (c >> 10),
        ((c & 0x3FF) << 6) | (d >> 8),
        ((d & 0xFF) << 8) | (e >> 6),
        ((e & 0x3F) << 10) | (f >> 4),
        ((f & 0xF) << 12) | (g >> 2),
        ((g & 0x3) << 14) | hh,
    ], 1).reshape(-1)
    return words


def colour_cast_frame(w: int, h: int, seed: int = 11, black: int = BLACK) -> np.ndarray:
    """What real footage looks like before white balance: R and B one to two EV below G (CFA gains .45/1/1/.3), a scene
    made of flat patches whose colour balance jumps by several EV at the patch borders, noise, a few clipped and a few
    black pixels.  Exercises the packed 16-bit medians of k_frame: inside a patch the colour differences stay within
    the 16-bit window around the local reference, across a border they leave it (32-bit fallback)."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    patch = ((xx // 24) * 7 + (yy // 20) * 13) % 5
    lum = 600 + 2500 * ((xx // 24 + yy // 20) % 3) + 3000 * np.sin(xx * 0.01) ** 2
    gains = np.array([[[0.45, 1.0], [1.0, 0.30]], [[1.6, 1.0], [1.0, 0.08]], [[0.06, 1.0], [1.0, 1.9]],
                      [[0.9, 1.0], [1.0, 0.9]], [[0.2, 1.0], [1.0, 0.2]]])
    v = lum * gains[patch, yy % 2, xx % 2] + rng.integers(-40, 41, (h, w))
    v = v + black
    v[rng.random((h, w)) < 0.002] = 16383
    v[rng.random((h, w)) < 0.002] = black
    return np.clip(v, 0, 16383).astype(np.uint16)


def low_light_frame(w: int, h: int, seed: int = 21, black: int = BLACK) -> np.ndarray:
    """Underexposed footage: a smooth scene 0..400 above black (R and B at 0.45 / 0.3 of G), Gaussian read noise of 7 DN, so a
    few per cent of the pixels of the dark half sit at or below the black level (ev = 0 / INT_MIN in raw2ev: the loader's
    out-of-table path), but no hard colour edges."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    lum = 8 + 400 * (xx / w) ** 2 + 60 * np.sin(yy * 0.02) ** 2
    gains = np.array([[0.45, 1.0], [1.0, 0.30]])
    v = lum * gains[yy % 2, xx % 2] + rng.normal(0.0, 7.0, (h, w)) + black
    return np.clip(np.rint(v), 0, 16383).astype(np.uint16)


def amaze_plane(w: int, h: int, seed: int = 1) -> np.ndarray:
    """Float RGGB plane in the scale the dual-ISO path hands to AMaZE (20-bit values, i.e. 0..16 after the
    tile loader's /65535): smooth gradients, a checker, clipped patches (> 0.8 * 65535 takes the
    Hamilton-Adams branches) and one-pixel textures (Nyquist branches)."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    base = 30000 + 25000 * np.sin(xx * 0.05) * np.cos(yy * 0.07) + 8000 * ((xx // 16 + yy // 16) & 1)
    raw = base * np.array([[0.6, 1.0], [1.0, 0.5]])[yy % 2, xx % 2] + rng.integers(-300, 300, (h, w))
    raw[10:20, 10:30] = 70000
    raw[h // 2:h // 2 + 12, w // 2:w // 2 + 40:2] = 500
    raw[h // 3:h // 3 + 9:2, 8:w - 8] += 9000
    return raw.clip(0, 0xFFFFF).astype(np.float32)


# ---------------------------------------------------------------- DNG header cases
HEADER_CAMERAS = ["Canon EOS 5D Mark III", "Canon EOS 5D Mark II", "Canon EOS 7D", "Canon EOS 6D", "Canon EOS 70D",
                  "Canon EOS 60D", "Canon EOS 50D", "Canon EOS 500D", "Canon EOS 550D", "Canon EOS 600D",
                  "Canon EOS 650D", "Canon EOS 700D", "Canon EOS 1100D", "Canon EOS M", "Canon EOS 100D", "X", "ab",
                  "Nikon D800 test body"]


def header_case(k: int):
    """Deterministic `frame_headers` for the header writer: case k -> (FrameHeaders, fps_override, basename).

    Cycles through every camera row (and unknown / very short names), every white-balance mode, crop / line-skipping /
    full-sensor geometries, frames with and without optical-black borders, zero and non-zero frame rates.
    """
    from . import abi
    state = [0]

    def r(n):
        state[0] += 1
        return int(_mix32((state[0] * 0x9E3779B1 + (k + 1) * 0x85EBCA6B) & _M32)) % n

    geoms = [(1920, 1080, 1920 + 146, 1080 + 28), (3584, 1320, 5936, 3950), (1728, 972, 1808, 1190), (1920, 672, 2080, 702),
             (1280, 434, 2080, 720), (640, 480, 720, 500), (2560, 1090, 2560, 1090)]
    w, h, rw, rh = geoms[k % len(geoms)]
    fh = abi.make_frame_headers(w, h, black=[2048, 1024, 2047, 0][r(4)], white=[15000, 16383, 60000][r(3)], guid=r(1 << 30),
                                raw_size=(rw, rh))
    ri = fh.rawi_hdr.raw_info
    x1, y1 = [(0, 0), (146, 28), (72, 26)][r(3)]
    ri.active_area[0], ri.active_area[1], ri.active_area[2], ri.active_area[3] = y1, x1, rh, rw   # y1, x1, y2, x2
    ri.crop[0], ri.crop[1], ri.crop[2], ri.crop[3] = r(200), r(100), w - r(16), h - r(16)
    ri.exposure_bias[0], ri.exposure_bias[1] = r(7) - 3, [0, 1, 2, 3][r(4)]
    name = HEADER_CAMERAS[k % len(HEADER_CAMERAS)].encode()
    for i, c in enumerate(name[:31]):
        fh.idnt_hdr.cameraName[i] = c
    fh.idnt_hdr.cameraModel = 0x80000285 + r(64)
    serial = ("%08X%08X%08X%08X" % (r(1 << 32), r(1 << 32), r(1 << 32), r(1 << 32))).encode()[: r(33)]
    for i, c in enumerate(serial):
        fh.idnt_hdr.cameraSerial[i] = c
    lens = [b"EF24-70mm f/2.8L II USM", b"", b"50", b"EF-S18-55mm f/3.5-5.6 IS STM kit"][r(4)][:31]
    for i, c in enumerate(lens):
        fh.lens_hdr.lensName[i] = c
    fh.lens_hdr.focalLength, fh.lens_hdr.focalDist, fh.lens_hdr.aperture = 10 + r(400), r(65535), 100 + r(2100)
    fh.expo_hdr.isoValue = [100, 1600, 25600, 102400][r(4)]
    fh.expo_hdr.shutterValue = [20000, 33333, 1000000, 250, 5_000_000_000][r(5)]
    fh.wbal_hdr.wb_mode = [0, 1, 2, 3, 4, 5, 6, 8, 9, 7][k % 10]
    fh.wbal_hdr.kelvin = 1700 + r(11000)
    fh.wbal_hdr.wbgain_r, fh.wbal_hdr.wbgain_g, fh.wbal_hdr.wbgain_b = 400 + r(800), 1024, 400 + r(1200)
    t = fh.rtci_hdr
    t.tm_sec, t.tm_min, t.tm_hour, t.tm_mday, t.tm_mon, t.tm_year = r(60), r(60), r(24), 1 + r(28), r(12), 110 + r(20)
    t.timestamp = r(1 << 20)
    fh.vidf_hdr.timestamp = t.timestamp + r(1 << 36)
    fh.vidf_hdr.frameNumber = [0, 1, 23, 24, 1439, 86400, 2_592_001][r(7)]
    nom, den = [(23976, 1000), (24000, 1001), (25000, 1000), (60000, 1000), (0, 0), (500, 1000), (30000, 0)][r(7)]
    fh.file_hdr.sourceFpsNom, fh.file_hdr.sourceFpsDenom = nom, den
    fps_override = [0.0, 0.0, 24.0, 23.976, 0.5][r(5)]
    base = [b"M27-1337", b"a", b"abc", b"/Volumes/CARD/DCIM/100EOS5D/M27-1337.MLV", b""][r(5)]
    return fh, fps_override, base


def fnv1a(a: np.ndarray) -> str:
    """64-bit FNV-1a over 64-bit words, 4096 lanes folded in parallel and then combined (fast enough for full-size
    frames): the hash of tests/golden/golden.json, shared by the tests and bench.py's output check."""
    b = np.ascontiguousarray(a).reshape(-1).view(np.uint8)
    pad = (-b.size) % 8
    if pad:
        b = np.concatenate([b, np.zeros(pad, np.uint8)])
    w = b.view(np.uint64)
    h = np.uint64(0xCBF29CE484222325)
    prime = np.uint64(0x100000001B3)
    lanes = 4096
    n = (w.size + lanes - 1) // lanes * lanes
    ww = np.zeros(n, np.uint64)
    ww[: w.size] = w
    ww = ww.reshape(-1, lanes)
    acc = np.full(lanes, h, np.uint64)
    with np.errstate(over="ignore"):
        for row in ww:
            acc = (acc ^ row) * prime
        out = h
        for v in acc:
            out = (out ^ v) * prime
    return f"{int(out):016x}"

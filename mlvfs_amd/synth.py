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
    if n_def:
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

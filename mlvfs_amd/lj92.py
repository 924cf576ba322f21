"""LJ92 (lossless JPEG) frame payloads -> 16-bit frames in HBM: ctypes face of csrc/lj92.cpp + csrc/k_lj92.hip
(reference mlvfs/main.c:617-681, mlvfs/lj92.c)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import lib


def info(stream: bytes) -> dict:
    L = lib.load()
    dims = (C.c_int * 4)()
    buf = np.frombuffer(stream, np.uint8)
    lib.check(L.mlvfs_amd_lj92_info(lib.ptr(buf), buf.size, dims), "lj92_info")
    return dict(width=dims[0], height=dims[1], bits=dims[2], predictor=dims[3])


def decode_frames(streams, xres: int, yres: int, out=None, torch_stream=None):
    """Decode a batch of JPEG streams (bytes-like, host memory) into a (n, yres, xres) int16 CUDA tensor (bit pattern of
    the uint16 pixels), untiled like main.c:646-667 does."""
    import torch
    L = lib.load()
    n = len(streams)
    bufs = [np.frombuffer(s, np.uint8) for s in streams]
    ptrs = (C.c_void_p * n)(*[b.ctypes.data for b in bufs])
    sizes = (C.c_size_t * n)(*[b.size for b in bufs])
    if out is None:
        out = torch.empty((n, yres, xres), dtype=torch.int16, device="cuda")
    st = C.c_void_p(torch_stream.cuda_stream) if torch_stream is not None else None
    lib.check(L.mlvfs_amd_lj92_decode_dev(ptrs, sizes, n, xres, yres, C.c_void_p(out.data_ptr()), out.stride(0) * 2, st), "lj92_decode_dev")
    return out


def encode(flat, w: int, h: int, bits: int = 14, read_len: int = 0, skip_len: int = 0, delin=None) -> bytes:
    """lj92_encode of the library (lj92.h:65-68): w x h values read from `flat` (host uint16) in runs of read_len values skip_len
    apart (0: contiguous) -> the JPEG stream.  Raises where the call refuses (what the reference cannot encode inside its arrays)."""
    L = lib.load()
    flat = np.ascontiguousarray(flat, np.uint16).reshape(-1)
    d = None if delin is None else np.ascontiguousarray(delin, np.uint16)
    enc = C.POINTER(C.c_uint8)()
    n = C.c_int(0)
    L.lj92_encode.argtypes = [C.c_void_p] + [C.c_int] * 5 + [C.c_void_p, C.c_int, C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_int)]
    rc = L.lj92_encode(flat.ctypes.data, w, h, bits, read_len or w * h, skip_len, None if d is None else d.ctypes.data,
                       0 if d is None else d.size, C.byref(enc), C.byref(n))
    if rc != 0:
        raise lib.MlvfsAmdError(f"lj92_encode failed ({rc}): {L.mlvfs_amd_last_error().decode()}")
    try:
        return C.string_at(enc, n.value)
    finally:
        libc = C.CDLL(None)
        libc.free.argtypes = [C.c_void_p]
        libc.free(enc)


def encode_table(hist, npix: int):
    """Host-only: the encoder's Huffman table for a class histogram (mlvfs_amd_lj92_encode_table), or None where it refuses."""
    L = lib.load()
    out = (C.c_int * 68)()
    hist = np.ascontiguousarray(hist, np.uint32)
    if L.mlvfs_amd_lj92_encode_table(hist.ctypes.data_as(C.POINTER(C.c_uint32)), npix, out) != 0:
        return None
    o = list(out)
    return dict(bits=o[0:16], nvalues=o[16], values=o[17:34], len=o[34:51], code=o[51:68])

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

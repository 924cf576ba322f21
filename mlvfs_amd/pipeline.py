"""Host-side mirror of MLVFS's per-frame orchestration.

`process_frame` reproduces the stage order and first-frame-of-clip logic of the
reference's process_frame (mlvfs/main.c:908-1005) on top of the DROP-IN symbols
of libmlvfs_amd.so -- i.e. it calls exactly what MLVFS's unchanged main.c would
call, with the same arguments, on host buffers:

    mlvfs_load_chunks                               main.c:923
    get_image_data -> dng_get_image_data            main.c:942 (687-700)
    [fix_pattern_noise]                             main.c:946-949
    [hdr_convert_data | cr2hdr20_convert_data]      main.c:951-959
    fix_focus_pixels; [fix_bad_pixels]              main.c:968-972
    [chroma_smooth]                                 main.c:975-978
    [stripes get/new/compute (first frame) + apply] main.c:980-997
    mlvfs_close_chunks                              main.c:998

The two chunk calls belong to MLVFS (resource_manager.c:285-317).  A MLVFS linked with integration/mlvfs_amd_wrap.c and
`-Wl,--wrap=mlvfs_load_chunks -Wl,--wrap=mlvfs_close_chunks` runs the library's frame bracket inside them; `WRAPPED = True`
makes this mirror do the same (MLVFS_AMD_PIPELINE_WRAPPED=1 in the environment, or set by a test; default: the plain link).

`MlvfsOptions` mirrors the flags of `struct mlvfs` (mlvfs/mlvfs.h:32-48) that reach
the hot path.  The reference's container IO / FUSE plumbing is out of scope: the
packed payload and the frame_headers are handed in by the caller.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass

import numpy as np

from . import abi, lib


WRAPPED = os.environ.get("MLVFS_AMD_PIPELINE_WRAPPED") == "1"


def mlvfs_load_chunks() -> None:
    """main.c:923 as the wrap shim extends it (integration/mlvfs_amd_wrap.c): arms the frame bracket after the real call."""
    if WRAPPED:
        lib.load().mlvfs_amd_frame_begin()


def mlvfs_close_chunks() -> None:
    """main.c:998 as the wrap shim extends it: ends the bracket (the frame is fetched here) before the real call."""
    if WRAPPED:
        lib.check(lib.load().mlvfs_amd_frame_end(), "frame_end")


@dataclass
class MlvfsOptions:
    chroma_smooth: int = 0          # 0, 2, 3, 5      (--cs2x2 / --cs3x3 / --cs5x5)
    fix_bad_pixels: int = 0         # 0, 1, 2         (--fix-bad-pixels / aggressive)
    fix_stripes: int = 0            # --fix-stripes
    dual_iso: int = 0               # 0, 1 preview, 2 full
    fix_pattern_noise: int = 0
    hdr_interpolation_method: int = 0
    hdr_no_fullres: int = 0
    hdr_no_alias_map: int = 0


def get_image_data(fh: abi.FrameHeaders, packed: np.ndarray, offset: int = 0, max_size: int | None = None) -> np.ndarray:
    """The uncompressed branch of get_image_data (main.c:684-703): unpack into a new buffer."""
    L = lib.load()
    size = L.dng_get_image_size(C.byref(fh)) if max_size is None else max_size
    out = np.zeros(size, np.uint8)
    packed = np.ascontiguousarray(packed, np.uint16)
    got = L.dng_get_image_data(C.byref(fh), lib.ptr(packed), lib.ptr(out), offset, size)
    if got != size:
        raise lib.MlvfsAmdError("dng_get_image_data failed: " + L.mlvfs_amd_last_error().decode())
    return out.view(np.uint16)


def process_frame(packed: np.ndarray, fh: abi.FrameHeaders, opt: MlvfsOptions, mlv_filename: str = "clip.MLV") -> np.ndarray:
    """One frame through the pipeline in the reference's order; returns the u16 image (h, w).

    `fh` is mutated like the reference mutates its frame_headers (dual-ISO levels)."""
    L = lib.load()
    w, h = fh.rawi_hdr.xRes, fh.rawi_hdr.yRes
    mlvfs_load_chunks()
    img = get_image_data(fh, packed).reshape(h, w)
    p = lib.ptr(img)
    if opt.fix_pattern_noise:
        L.fix_pattern_noise(p, w, h, fh.rawi_hdr.raw_info.white_level, 0)
    is_dual_iso = 0
    if opt.dual_iso == 1:
        is_dual_iso = L.hdr_convert_data(C.byref(fh), p, 0, img.nbytes)
    elif opt.dual_iso == 2:
        is_dual_iso = L.cr2hdr20_convert_data(C.byref(fh), p, opt.hdr_interpolation_method, int(not opt.hdr_no_fullres),
                                              int(not opt.hdr_no_alias_map), opt.chroma_smooth, opt.fix_bad_pixels)
    if not is_dual_iso:
        L.fix_focus_pixels(C.byref(fh), p, 0)
        if opt.fix_bad_pixels:
            L.fix_bad_pixels(C.byref(fh), p, int(opt.fix_bad_pixels == 2), is_dual_iso)
    if opt.chroma_smooth and opt.dual_iso != 2:
        L.chroma_smooth(C.byref(fh), p, opt.chroma_smooth)
    if opt.fix_stripes:
        name = mlv_filename.encode()
        corr = L.stripes_get_correction(name)
        if not corr:
            corr = L.stripes_new_correction(name)
            if corr:
                L.stripes_compute_correction(C.byref(fh), corr, p, 0, img.size)
        L.stripes_apply_correction(C.byref(fh), corr, p, 0, img.size)
    mlvfs_close_chunks()
    return img

"""Device-resident frame streams on top of the mlvfs_amd_* C ABI.

torch is plumbing here: it owns the HBM buffers and the HIP stream, nothing else.
All arithmetic happens in libmlvfs_amd.so's HIP kernels; device pointers are
handed to the C ABI as plain integers.

`ClipStream` is the throughput form of process_frame (mlvfs/main.c:908-1005) for
one clip whose packed payloads are already in HBM:

  * first frame of the clip  (main.c:969-988): unpack -> detect bad pixels ->
    repair -> chroma smooth -> stripes histogram/coefficients, which fixes the two
    per-clip artefacts (pixel map, 8 coefficients);
  * every frame: ONE fused launch sequence (pixel repair values + fused kernel)
    over the whole batch.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import lib, synth


def _dptr(t: torch.Tensor) -> C.c_void_p:
    return C.c_void_p(t.data_ptr())


def _stream_ptr() -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class ClipStream:
    def __init__(self, width: int, height: int, bpp: int = 14, black: int = synth.BLACK, white: int = synth.WHITE,
                 device: int = 0, pan=(0, 0)):
        self.L = lib.load()
        self.w, self.h, self.bpp, self.black, self.white = width, height, bpp, black, white
        self.device = device
        torch.cuda.set_device(device)
        lib.check(self.L.mlvfs_amd_init(device), "mlvfs_amd_init")
        self.geom = lib.Geom(width, height, bpp, black, white, pan[0], pan[1])
        self.clip = self.L.mlvfs_amd_clip_create(C.byref(self.geom))
        if not self.clip:
            raise lib.MlvfsAmdError(self.L.mlvfs_amd_last_error().decode())
        self.npix = width * height
        # frame strides: packed payload rounded up to 16 B (the reference reads one pixel past the end)
        self.packed_stride = ((self.npix * bpp + 7) // 8 + 2 + 15) // 16 * 16
        self.out_stride = self.npix * 2
        self.frame_size = self.npix * bpp // 8

    def close(self):
        if self.clip:
            self.L.mlvfs_amd_clip_destroy(self.clip)
            self.clip = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ buffers
    def alloc_packed(self, nframes: int) -> torch.Tensor:
        return torch.zeros((nframes, self.packed_stride), dtype=torch.uint8, device=f"cuda:{self.device}")

    def alloc_out(self, nframes: int) -> torch.Tensor:
        return torch.empty((nframes, self.h, self.w), dtype=torch.int16, device=f"cuda:{self.device}")

    def upload_packed(self, packed_frames) -> torch.Tensor:
        """numpy uint16 word arrays (one per frame) -> device stream buffer."""
        buf = np.zeros((len(packed_frames), self.packed_stride), np.uint8)
        for i, p in enumerate(packed_frames):
            b = np.ascontiguousarray(p, "<u2").view(np.uint8)
            n = min(b.size, self.packed_stride)
            buf[i, :n] = b[:n]
        return torch.from_numpy(buf).to(f"cuda:{self.device}")

    def synth_packed(self, nframes: int, seed: int = 1, first_frame: int = 0, kind: str = "normal") -> torch.Tensor:
        """Build a synthetic packed 14-bit stream directly in HBM (bench.py)."""
        assert self.bpp == 14 and self.npix % 8 == 0
        dev = torch.zeros(1, device=f"cuda:{self.device}")
        out = self.alloc_packed(nframes)
        gen = getattr(synth, kind + "_frame")
        for f in range(nframes):
            frame = gen(self.w, self.h, seed=seed, frame=first_frame + f, black=self.black, like=dev)
            words = synth.pack14(frame).to(torch.int32)
            lo = (words & 0xFF).to(torch.uint8)
            hi = (words >> 8).to(torch.uint8)
            out[f, : 2 * words.numel()] = torch.stack([lo, hi], 1).reshape(-1)
        return out

    # ------------------------------------------------------------------ per-clip state
    def set_stripes(self, needed: int, coeffs) -> None:
        co = np.ascontiguousarray(coeffs, np.int32)
        lib.check(self.L.mlvfs_amd_clip_set_stripes(self.clip, int(needed), lib.ptr(co)))

    def set_t16_layout(self, layout: int) -> None:
        """0 = auto (from the first processed frame), 1 = plain, 2 = spread (dark footage); results are identical."""
        lib.check(self.L.mlvfs_amd_clip_set_t16_layout(self.clip, layout), "clip_set_t16_layout")

    def get_t16_layout(self) -> int:
        return int(self.L.mlvfs_amd_clip_get_t16_layout(self.clip))

    def get_stripes(self):
        co = np.zeros(8, np.int32)
        needed = C.c_int(0)
        self.L.mlvfs_amd_clip_get_stripes(self.clip, C.byref(needed), lib.ptr(co))
        return needed.value, co

    def set_pixel_map(self, xy, kind: int = 0, dual_iso: int = 0) -> None:
        xy = np.ascontiguousarray(xy, np.int32).reshape(-1)
        lib.check(self.L.mlvfs_amd_clip_set_pixel_map(self.clip, lib.ptr(xy) if xy.size else None, xy.size // 2,
                                                      kind, dual_iso), "set_pixel_map")

    def get_pixel_map(self) -> np.ndarray:
        n = self.L.mlvfs_amd_clip_get_pixel_map(self.clip, None, 0)
        xy = np.zeros((max(n, 1), 2), np.int32)
        self.L.mlvfs_amd_clip_get_pixel_map(self.clip, lib.ptr(xy), n)
        return xy[:n]

    # ------------------------------------------------------------------ stages
    def unpack(self, packed: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
        n = packed.shape[0]
        out = self.alloc_out(n) if out is None else out
        lib.check(self.L.mlvfs_amd_unpack_dev(C.byref(self.geom), _dptr(packed), self.packed_stride, _dptr(out),
                                              self.out_stride, n, _stream_ptr()), "unpack_dev")
        return out

    def chroma_smooth(self, frames: torch.Tensor, method: int) -> torch.Tensor:
        out = torch.empty_like(frames)
        lib.check(self.L.mlvfs_amd_chroma_smooth_dev(C.byref(self.geom), _dptr(frames), _dptr(out), self.out_stride,
                                                     method, frames.shape[0], _stream_ptr()), "chroma_smooth_dev")
        return out

    def detect_bad_pixels(self, frame: torch.Tensor, aggressive: int = 0) -> np.ndarray:
        lib.check(self.L.mlvfs_amd_detect_bad_pixels_dev(self.clip, _dptr(frame), aggressive, _stream_ptr()),
                  "detect_bad_pixels_dev")
        return self.get_pixel_map()

    def fix_pixels(self, frames: torch.Tensor) -> torch.Tensor:
        lib.check(self.L.mlvfs_amd_fix_pixels_dev(self.clip, _dptr(frames), self.out_stride, frames.shape[0],
                                                  _stream_ptr()), "fix_pixels_dev")
        return frames

    def stripes_compute(self, frame: torch.Tensor, rand_mode: int = 1):
        lib.check(self.L.mlvfs_amd_stripes_compute_dev(self.clip, _dptr(frame), self.frame_size, rand_mode,
                                                       _stream_ptr()), "stripes_compute_dev")
        return self.get_stripes()

    def stripes_apply(self, frames: torch.Tensor) -> torch.Tensor:
        lib.check(self.L.mlvfs_amd_stripes_apply_dev(self.clip, _dptr(frames), self.out_stride, frames.shape[0],
                                                     _stream_ptr()), "stripes_apply_dev")
        return frames

    def process(self, packed: torch.Tensor, out: torch.Tensor | None = None, cs: int = 0, fix_pixels: bool = False,
                stripes: bool = False) -> torch.Tensor:
        """Fused steady-state pass over a batch of packed frames (clip state must be set)."""
        n = packed.shape[0]
        out = self.alloc_out(n) if out is None else out
        lib.check(self.L.mlvfs_amd_process_frames_dev(self.clip, _dptr(packed), self.packed_stride, _dptr(out),
                                                      self.out_stride, n, cs, int(fix_pixels), int(stripes),
                                                      _stream_ptr()), "process_frames_dev")
        return out

    def process_unpacked(self, frames: torch.Tensor, out: torch.Tensor | None = None, cs: int = 0, fix_pixels: bool = False,
                         stripes: bool = False) -> torch.Tensor:
        """The same pass for frames that are already 16-bit in HBM (decoded LJ92 payloads)."""
        n = frames.shape[0]
        out = self.alloc_out(n) if out is None else out
        lib.check(self.L.mlvfs_amd_process_unpacked_dev(self.clip, _dptr(frames), self.out_stride, _dptr(out), self.out_stride,
                                                        n, cs, int(fix_pixels), int(stripes), _stream_ptr()),
                  "process_unpacked_dev")
        return out

    def process_host(self, packed: torch.Tensor, out: torch.Tensor | None = None, cs: int = 0, fix_pixels: bool = False,
                     stripes: bool = False, chunk: int = 8) -> torch.Tensor:
        """The same pass for frames in HOST memory (uint8 tensors, ideally pinned): chunked and triple-buffered over PCIe."""
        assert packed.device.type == "cpu" and packed.dtype == torch.uint8 and packed.is_contiguous()
        n = packed.shape[0]
        if out is None:
            out = torch.empty((n, self.out_stride), dtype=torch.uint8, pin_memory=True)
        lib.check(self.L.mlvfs_amd_process_frames_host(self.clip, C.c_void_p(packed.data_ptr()), packed.stride(0),
                                                       C.c_void_p(out.data_ptr()), out.stride(0), n, cs, int(fix_pixels),
                                                       int(stripes), chunk), "process_frames_host")
        return out

    # ------------------------------------------------------------------ first frame of a clip
    def analyse_first_frame(self, packed0: torch.Tensor, cs: int = 0, bad_pix: int = 0, stripes: bool = False,
                            rand_mode: int = 1) -> torch.Tensor:
        """main.c:942-988 for the first processed frame: fixes the clip's pixel map and
        stripe coefficients and returns that frame fully processed."""
        frame = self.unpack(packed0[:1])
        if bad_pix:
            self.detect_bad_pixels(frame[0], int(bad_pix == 2))
            self.fix_pixels(frame)
        if cs:
            frame = self.chroma_smooth(frame, cs)
        if stripes:
            self.stripes_compute(frame[0], rand_mode)
            self.stripes_apply(frame)
        return frame


def to_numpy_u16(t: torch.Tensor) -> np.ndarray:
    return t.detach().cpu().numpy().view(np.uint16)

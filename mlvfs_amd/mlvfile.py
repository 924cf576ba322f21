"""MLV v2.0 containers: a writer for synthetic clips (tests, tools) and the ctypes face of the reader in
libmlvfs_amd.so (csrc/mlvreader.cpp).

Block layouts are the public Magic Lantern MLV v2.0 format as declared in include/mlvfs_abi.h (mirrors of
mlvfs/mlv.h:40-239).  Every block starts with a 4-byte tag, its 32-bit size and a 64-bit timestamp in microseconds;
the file header MLVI has the format's version string where the timestamp would be.
"""
from __future__ import annotations

import ctypes as C
import os
import struct

import numpy as np

from . import abi, lib


def _blob(s) -> bytes:
    return bytes(s)


def block(hdr, tag: bytes, timestamp: int, payload: bytes = b"", pad: int = 0) -> bytes:
    """Serialise one block: `hdr` is an abi.*Hdr instance whose prefix is filled in here."""
    body = payload + b"\0" * pad
    hdr.blockType[:] = tag
    hdr.blockSize = C.sizeof(hdr) + len(body)
    hdr.timestamp = timestamp
    return _blob(hdr) + body


def file_header(guid: int, file_num: int, file_count: int, frames: int, fps=(24000, 1001), video_class: int = 1) -> bytes:
    h = abi.FileHdr()
    h.fileMagic[:] = b"MLVI"
    h.blockSize = C.sizeof(h)
    h.versionString[:] = b"v2.0\0\0\0\0"
    h.fileGuid, h.fileNum, h.fileCount = guid, file_num, file_count
    h.videoClass, h.audioClass = video_class, 0
    h.videoFrameCount, h.audioFrameCount = frames, 0
    h.sourceFpsNom, h.sourceFpsDenom = fps
    return _blob(h)


def raw_block(tag: bytes, timestamp: int, payload: bytes) -> bytes:
    """A block this library has no struct for (INFO, NULL, AUDF, DEBG ...): tag, size, timestamp, payload."""
    return tag + struct.pack("<IQ", 16 + len(payload), timestamp) + payload


def write_clip(path: str, payloads, w: int, h: int, bpp: int = 14, black: int = 2048, white: int = 15000, *, guid: int = 0x1234ABCD,
               chunks: int = 1, frame_space: int = 0, shuffle: bool = False, extras: bool = True, camera: bytes = b"Canon EOS 5D Mark III",
               video_class: int = 1) -> list[str]:
    """Write `payloads` (bytes per frame, as stored on the card) as <path> (+ .M00 ... for chunks > 1).

    Frame k goes to chunk k % chunks.  extras: EXPO / LENS blocks that change in the middle of the clip, a NULL padding
    block, an INFO block and an audio frame, so that the index has to skip, order and classify them.  shuffle: blocks of a
    chunk are written out of timestamp order (as a camera does when its buffers drain unevenly).  Returns the file names.
    """
    n = len(payloads)
    names = [path] + [path[:-2] + "%02d" % i for i in range(chunks - 1)]
    per_chunk: list[list[bytes]] = [[] for _ in range(chunks)]
    for c in range(chunks):
        per_chunk[c].append(file_header(guid, c, chunks, len(range(c, n, chunks)), video_class=video_class))
    t = 1000
    rawi = abi.RawiHdr()
    rawi.xRes, rawi.yRes = w, h
    ri = rawi.raw_info
    ri.api_version, ri.height, ri.width, ri.pitch = 1, h + 28, w + 146, (w + 146) * bpp // 8
    ri.frame_size, ri.bits_per_pixel, ri.black_level, ri.white_level = w * h * bpp // 8, bpp, black, white
    ri.active_area[0], ri.active_area[1], ri.active_area[2], ri.active_area[3] = 28, 146, h + 28, w + 146
    ri.cfa_pattern, ri.exposure_bias[1], ri.dynamic_range = 0x02010100, 1, 1100
    per_chunk[0].append(block(rawi, b"RAWI", t)); t += 7
    idnt = abi.IdntHdr()
    idnt.cameraName[:len(camera)] = camera
    idnt.cameraModel = 0x80000285
    idnt.cameraSerial[:8] = b"0123ABCD"
    per_chunk[0].append(block(idnt, b"IDNT", t)); t += 7
    rtci = abi.RtciHdr()
    rtci.tm_sec, rtci.tm_min, rtci.tm_hour, rtci.tm_mday, rtci.tm_mon, rtci.tm_year = 30, 59, 23, 3, 9, 126
    per_chunk[0].append(block(rtci, b"RTCI", t)); t += 7

    def expo(iso, at):
        e = abi.ExpoHdr()
        e.isoMode, e.isoValue, e.isoAnalog, e.digitalGain, e.shutterValue = 0, iso, iso, 0, 20000
        return block(e, b"EXPO", at)

    def lens(focal, at):
        le = abi.LensHdr()
        le.focalLength, le.focalDist, le.aperture = focal, 65535, 280
        le.lensName[:23] = b"EF24-70mm f/2.8L II USM"
        return block(le, b"LENS", at)

    per_chunk[0].append(expo(100, t)); t += 7
    per_chunk[0].append(lens(24, t)); t += 7
    wb = abi.WbalHdr()
    wb.wb_mode, wb.kelvin, wb.wbgain_r, wb.wbgain_g, wb.wbgain_b = 9, 5200, 2000, 1024, 1600
    per_chunk[0].append(block(wb, b"WBAL", t)); t += 7
    if extras:
        per_chunk[0].append(raw_block(b"INFO", t, b"take 1\0\0")); t += 7
    for k, p in enumerate(payloads):
        c = k % chunks
        if extras and k == n // 2:
            per_chunk[c].append(expo(800, t)); t += 3                  # the second half of the clip is at ISO 800, 35 mm
            per_chunk[c].append(lens(35, t)); t += 3
        if extras and k % 5 == 2:
            per_chunk[c].append(raw_block(b"NULL", t, b"\0" * 48))        # padding blocks are not indexed
        if extras and k % 4 == 1:
            per_chunk[c].append(raw_block(b"AUDF", t + 1, struct.pack("<II", k, 0) + b"\x55" * 64))
        v = abi.VidfHdr()
        v.frameNumber, v.frameSpace = k, frame_space
        per_chunk[c].append(block(v, b"VIDF", t + 2, b"\xEE" * frame_space + bytes(p)))
        t += 41708
    if shuffle:
        for c in range(chunks):
            body = per_chunk[c][1:]
            for i in range(0, len(body) - 1, 3):                        # swap neighbours: locally out of order
                body[i], body[i + 1] = body[i + 1], body[i]
            per_chunk[c][1:] = body
    for name, blocks in zip(names, per_chunk):
        with open(name, "wb") as f:
            f.write(b"".join(blocks))
    return names


class MlvReader:
    """csrc/mlvreader.cpp through ctypes: index, per-frame headers, payload reads, file -> GPU pipeline."""

    def __init__(self, path: str, use_idx_file: bool = False):
        self.L = lib.load()
        self.h = self.L.mlvfs_amd_mlv_open(os.fsencode(path), int(use_idx_file))
        if not self.h:
            raise lib.MlvfsAmdError("mlvfs_amd_mlv_open failed: " + self.L.mlvfs_amd_last_error().decode())

    def close(self):
        if self.h:
            self.L.mlvfs_amd_mlv_close(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    @property
    def frame_count(self) -> int:
        return self.L.mlvfs_amd_mlv_frame_count(self.h)

    @property
    def chunk_count(self) -> int:
        return self.L.mlvfs_amd_mlv_chunk_count(self.h)

    def xref(self) -> bytes:
        n = self.L.mlvfs_amd_mlv_xref(self.h, None, 0)
        buf = np.zeros(n, np.uint8)
        self.L.mlvfs_amd_mlv_xref(self.h, lib.ptr(buf), n)
        return buf.tobytes()

    def frame_headers(self, index: int):
        fh = abi.FrameHeaders()
        ok = self.L.mlvfs_amd_mlv_frame_headers(self.h, index, C.byref(fh))
        return ok, fh

    def read_frames(self, first: int, count: int, stride: int, io_threads: int = 4) -> np.ndarray:
        out = np.zeros((count, stride), np.uint8)
        lib.check(self.L.mlvfs_amd_mlv_read_frames(self.h, first, count, lib.ptr(out), stride, io_threads), "mlv_read_frames")
        return out

    def process(self, clip, first: int, count: int, out: np.ndarray, cs: int, fix_pixels: bool, stripes: bool, batch: int = 0,
                io_threads: int = 0) -> np.ndarray:
        """file -> fused GPU pipeline -> `out` (count x h x w uint16, C-contiguous); clip = mlvfs_amd_clip_t handle."""
        stride = out.strides[0]
        lib.check(self.L.mlvfs_amd_mlv_process(self.h, clip, first, count, lib.ptr(out), stride, cs, int(fix_pixels), int(stripes),
                                               batch, io_threads), "mlv_process")
        return out

    def process_dualiso(self, first: int, count: int, out: np.ndarray, interp: int = 0, fullres: int = 1, alias_map: int = 1, cs: int = 0,
                        batch: int = 0, io_threads: int = 0) -> np.ndarray:
        """file -> unpack -> batched full dual-ISO conversion -> `out` (count x h x w uint16); returns the per-frame results (1 converted)."""
        res = np.zeros(count, np.int32)
        lib.check(self.L.mlvfs_amd_mlv_process_dualiso(self.h, first, count, lib.ptr(out), out.strides[0], interp, fullres, alias_map, cs, batch,
                                                       io_threads, lib.ptr(res)), "mlv_process_dualiso")
        return res

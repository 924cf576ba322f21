"""`struct frame_headers` layout: include/mlvfs_abi.h == mlvfs_amd.abi (ctypes) ==
the reference's own headers (mlvfs/mlvfs.h:51-63, mlv.h, raw.h) where available."""
import ctypes as C
import os
import subprocess

import pytest

from mlvfs_amd import abi

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference/mlvfs"

FROZEN = {  # offsets measured from the reference headers (x86-64)
    "sizeof_frame_headers": 592, "sizeof_raw_info": 160, "fileNumber": 0, "position": 8, "vidf_hdr": 16,
    "vidf_hdr.panPosX": 40, "vidf_hdr.panPosY": 42, "file_hdr": 48, "file_hdr.fileGuid": 64, "rtci_hdr": 100,
    "idnt_hdr": 144, "idnt_hdr.cameraModel": 192, "rawi_hdr": 228, "rawi_hdr.xRes": 244, "rawi_hdr.yRes": 246,
    "rawi_hdr.raw_info": 248, "rawi_hdr.raw_info.height": 256, "rawi_hdr.raw_info.width": 260,
    "rawi_hdr.raw_info.frame_size": 268, "rawi_hdr.raw_info.bits_per_pixel": 272, "rawi_hdr.raw_info.black_level": 276,
    "rawi_hdr.raw_info.white_level": 280, "expo_hdr": 408, "lens_hdr": 448, "wbal_hdr": 544,
}


def probe(tmp_path, flags):
    exe = tmp_path / "abi_probe"
    subprocess.run(["gcc", "-std=gnu99", *flags, os.path.join(HERE, "abi_probe.c"), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout
    return {k: int(v) for k, v in (line.split() for line in out.strip().splitlines())}


def test_header_matches_frozen_layout(tmp_path):
    mine = probe(tmp_path, ["-I", os.path.join(ROOT, "include")])
    for k, v in FROZEN.items():
        assert mine[k] == v, k


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present")
def test_header_matches_reference_headers(tmp_path):
    mine = probe(tmp_path, ["-I", os.path.join(ROOT, "include")])
    ref = probe(tmp_path, ["-DUSE_REFERENCE", "-I", REF])
    assert mine == ref


def test_ctypes_mirror():
    assert C.sizeof(abi.FrameHeaders) == FROZEN["sizeof_frame_headers"]
    assert C.sizeof(abi.RawInfo) == FROZEN["sizeof_raw_info"]
    FH = abi.FrameHeaders
    assert FH.vidf_hdr.offset == 16 and FH.file_hdr.offset == 48 and FH.rtci_hdr.offset == 100
    assert FH.idnt_hdr.offset == 144 and FH.rawi_hdr.offset == 228 and FH.expo_hdr.offset == 408
    assert FH.lens_hdr.offset == 448 and FH.wbal_hdr.offset == 544
    assert abi.RawiHdr.raw_info.offset == 20 and abi.RawInfo.black_level.offset == 28
    assert abi.VidfHdr.panPosX.offset == 24 and abi.FileHdr.fileGuid.offset == 16 and abi.IdntHdr.cameraModel.offset == 48
    fh = abi.make_frame_headers(3584, 1320, guid=5, pan=(8, 2), camera=0x80000331)
    assert fh.rawi_hdr.raw_info.frame_size == 3584 * 1320 * 14 // 8 and fh.file_hdr.fileGuid == 5

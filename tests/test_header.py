"""DNG header writer `dng_get_header_data` (reference mlvfs/dng.c:597-789).

Host-only code of libmlvfs_amd.so, so these run without a GPU: against the committed vectors made by the
reference (tests/golden/header_cases.npz), against the reference build itself where oracle/_ref exists, and
structurally (a TIFF walk of what was written)."""
import ctypes as C
import os
import struct

import numpy as np
import pytest

from mlvfs_amd import abi, lib, synth

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "header_cases.npz"))


def run(L, blob, fps, base, offset=0, max_size=65536):
    fh = abi.FrameHeaders.from_buffer_copy(bytes(bytearray(blob)))
    out = np.full(max(max_size, 1) + 8, 0xA5, np.uint8)                     # 8 guard bytes behind the request
    n = L.dng_get_header_data(C.byref(fh), lib.ptr(out), offset, max_size, float(fps), base)
    assert (out[max_size:] == 0xA5).all(), "wrote past max_size"
    return n, out[:max_size], np.frombuffer(bytes(fh), np.uint8)


def test_case_generator_matches_fixture():
    for k in (0, 17, 89):
        fh, fps, base = synth.header_case(k)
        assert bytes(fh) == GOLD["blob_in"][k].tobytes() and fps == GOLD["fps"][k] and base == GOLD["base"][k]


@pytest.mark.parametrize("k", range(len(GOLD["nret"])))
def test_header_equals_reference_vector(amd, k):
    n, out, after = run(amd, GOLD["blob_in"][k], GOLD["fps"][k], bytes(GOLD["base"][k]))
    keep = GOLD["head"].shape[1]
    assert n == GOLD["nret"][k] == 65536
    assert np.array_equal(out[:keep], GOLD["head"][k]) and not out[keep:].any()
    assert np.array_equal(after, GOLD["blob_out"][k])                         # active-area rewrite (dng.c:665-672)


def test_header_equals_reference_build(amd, reference):
    """300 further cases plus partial reads, byte for byte against the reference's own function."""
    for k in range(90, 390):
        fh, fps, base = synth.header_case(k)
        blob = np.frombuffer(bytes(fh), np.uint8)
        n0, want, after0 = reference.header_data(blob, 0, 65536, fps, base)
        n1, got, after1 = run(amd, blob, fps, base)
        assert n0 == n1 and np.array_equal(got, want) and np.array_equal(after0, after1), k
    fh, fps, base = synth.header_case(3)
    blob = np.frombuffer(bytes(fh), np.uint8)
    for offset, size in ((0, 0), (0, 1), (0, 700), (8, 512), (644, 900), (65000, 536), (100, 65436), (0, 9461760 // 144)):
        n0, want, _ = reference.header_data(blob, offset, size, fps, base)     # within the header: defined behaviour
        n1, got, _ = run(amd, blob, fps, base, offset, size)
        assert n0 == n1 == min(size, 65536) and np.array_equal(got[:n1], want[:n0]), (offset, size)


def test_size_rule_and_reads_past_the_end(amd):
    """returned size = min(max_size, 65536) whatever the offset (dng.c:779); bytes past the header read as zero."""
    fh, fps, base = synth.header_case(5)
    blob = np.frombuffer(bytes(fh), np.uint8)
    _, full, _ = run(amd, blob, fps, base)
    n, got, _ = run(amd, blob, fps, base, offset=65000, max_size=4096)
    assert n == 4096 and np.array_equal(got[:536], full[65000:]) and not got[536:].any()
    n, got, _ = run(amd, blob, fps, base, offset=70000, max_size=100)
    assert n == 100 and not got.any()
    n, got, _ = run(amd, blob, fps, base, offset=0, max_size=100000)
    assert n == 65536 and np.array_equal(got[:65536], full)


def walk_ifd(buf, at):
    (count,) = struct.unpack_from("<H", buf, at)
    ent = [struct.unpack_from("<HHII", buf, at + 2 + 12 * i) for i in range(count)]
    (nxt,) = struct.unpack_from("<I", buf, at + 2 + 12 * count)
    return ent, nxt


def test_header_is_a_wellformed_tiff(amd):
    size = {1: 1, 2: 1, 3: 2, 4: 4, 5: 8, 7: 1, 10: 8}
    for k in range(0, 60, 7):
        fh, fps, base = synth.header_case(k)
        _, out, _ = run(amd, np.frombuffer(bytes(fh), np.uint8), fps, base)
        buf = out.tobytes()
        assert struct.unpack_from("<HHI", buf, 0) == (0x4949, 42, 8)
        ifd0, nxt = walk_ifd(buf, 8)
        assert len(ifd0) == 41 and nxt == 0
        tags = [e[0] for e in ifd0]
        assert tags == sorted(tags) or tags[-4:] == [51043, 51044, 51081, 51109]
        d = {e[0]: e for e in ifd0}
        assert d[256][3] == fh.rawi_hdr.xRes and d[257][3] == fh.rawi_hdr.yRes and d[273][3] == 65536
        assert d[279][3] == fh.rawi_hdr.xRes * fh.rawi_hdr.yRes * 2 and d[50714][3] == fh.rawi_hdr.raw_info.black_level
        exif, nxt = walk_ifd(buf, d[34665][3])
        assert len(exif) == 11 and nxt == 0
        ends = []
        for tag, typ, cnt, val in ifd0 + exif:
            nbytes = size[typ] * cnt
            if nbytes > 4:
                assert 8 + 2 + 41 * 12 + 4 + 2 + 11 * 12 + 4 <= val and val + nbytes <= 65536, tag
                ends.append((val, val + nbytes))
        ends.sort()
        assert all(a[1] <= b[0] for a, b in zip(ends, ends[1:])), "out-of-line values overlap"
        model = bytes(fh.idnt_hdr.cameraName).split(b"\0")[0]
        off, n = d[272][3], d[272][2]
        assert n == len(model) + 1 and (buf[off:off + n] == model + b"\0" if n > 4 else struct.pack("<I", off)[:n] == model + b"\0")

"""LZMA payloads (main.c:598-616 -> LZMA/LzmaLib.c) and the animated GIF preview (8f N4; gif.c:82-244).

CPU: the library's LZMA decoder (host code) against the reference's vendored decoder on streams made by the reference's
encoder (oracle/_ref, built from /root/reference by oracle/Makefile) and against committed vectors (tests/golden/lzma_vectors.npz,
made by tests/golden/make_lzma_gif_golden.py) -- property bytes, dictionary sizes, truncation, corruption, end markers.
GPU: LZMA clips through the container reader and the fused pipeline; the GIF file byte for byte against the reference's gif.c
(its frame fetch is the reference's get_image_data, sliced out of main.c at build time) and against committed hashes."""
import ctypes as C
import hashlib
import os

import numpy as np
import pytest

from mlvfs_amd import lib, mlvfile, synth

HERE = os.path.dirname(os.path.abspath(__file__))
W, H = 128, 64


def uncompress(amd, payload: bytes):
    want = int.from_bytes(payload[:4], "little")
    out = np.zeros(max(want, 1), np.uint8)
    n = C.c_size_t(0)
    p = np.frombuffer(payload, np.uint8).copy()
    rc = amd.mlvfs_amd_lzma_uncompress(lib.ptr(p), p.size, lib.ptr(out), want, C.byref(n))
    return rc, out[:n.value].tobytes()


def sample_data(kind: int) -> bytes:
    rng = np.random.default_rng(100 + kind)
    if kind == 0:
        return synth.pack_bits(synth.normal_frame(W, H, seed=4)).tobytes()                       # a packed frame, the real use
    if kind == 1:
        return bytes(rng.integers(0, 256, 70000, dtype=np.uint8))                                # incompressible
    if kind == 2:
        return b"abcabcabd" * 9000 + bytes(5000) + b"x"                                          # long matches, long runs, repeats
    if kind == 3:
        return bytes(rng.integers(0, 4, 50000, dtype=np.uint8))                                  # small alphabet
    if kind == 4:
        return b"q"                                                                              # one literal
    return (b"The quick brown fox jumps over the lazy dog. " * 400)[:17001]


PROPS = [(5, 1 << 20, 3, 0, 2), (1, 1 << 16, 0, 0, 0), (9, 1 << 22, 8, 4, 4), (5, 4096, 4, 2, 1), (3, 12288, 1, 1, 3)]


@pytest.mark.parametrize("kind", range(6))
@pytest.mark.parametrize("props", range(len(PROPS)))
def test_lzma_decoder_equals_the_reference(amd, reference, kind, props):
    data = sample_data(kind)
    payload = reference.lzma_payload(data, *PROPS[props])
    r0, want = reference.lzma_uncompress(payload)
    r1, got = uncompress(amd, payload)
    assert r0 == r1 == 0 and got == want == data


def test_lzma_damaged_streams_fail_like_the_reference(amd, reference):
    """Truncated input, corrupted bytes, a size word larger / smaller than the stream holds: same return code and (for return
    code 0) the same bytes as LzmaUncompress."""
    rng = np.random.default_rng(7)
    data = sample_data(0)
    payload = reference.lzma_payload(data)
    cases = [payload[:9], payload[:10], payload[:13], payload[:14], payload[:200], payload[:-1], payload[:-5], payload[:-6]]
    for cut in (20, 21, 22, 23, 24, 25, 40, 1000):
        cases.append(payload[:len(payload) - cut])
    for _ in range(60):                                               # flipped bytes anywhere behind the size word
        b = bytearray(payload)
        k = int(rng.integers(4, len(b)))
        b[k] ^= int(rng.integers(1, 256))
        cases.append(bytes(b))
    small = bytearray(payload); small[:4] = (len(data) // 2).to_bytes(4, "little"); cases.append(bytes(small))       # output fills first
    large = bytearray(payload); large[:4] = (len(data) + 100).to_bytes(4, "little"); cases.append(bytes(large))     # input ends first
    ok = bad = 0
    for c in cases:
        r0, want = reference.lzma_uncompress(c)
        r1, got = uncompress(amd, c)
        assert r0 == r1, (len(c), r0, r1)
        if r0 == 0:
            assert got == want
            ok += 1
        else:
            bad += 1
    assert ok >= 2 and bad >= 10


def test_lzma_committed_vectors(amd):
    """streams of the reference's encoder with the outputs of the reference's decoder, committed as data"""
    g = np.load(os.path.join(HERE, "golden", "lzma_vectors.npz"))
    n = int(g["count"])
    assert n >= 12
    for k in range(n):
        rc, got = uncompress(amd, g[f"payload{k}"].tobytes())
        assert rc == int(g[f"rc{k}"])
        if rc == 0:
            assert hashlib.sha256(got).hexdigest() == str(g[f"sha{k}"])


# ------------------------------------------------------------------ clips
def lzma_clip(reference, tmp_path, n=6, w=W, h=H, **kw):
    frames = [synth.normal_frame(w, h, seed=3, frame=k) for k in range(n)]
    payloads = [reference.lzma_payload(synth.pack_bits(f).tobytes()) for f in frames]
    names = mlvfile.write_clip(str(tmp_path / "M01-0001.MLV"), payloads, w, h, video_class=1 | 0x80, **kw)
    return names, frames


def test_lzma_clip_reads_like_an_uncompressed_one(reference, tmp_path):
    names, frames = lzma_clip(reference, tmp_path, chunks=2)
    with mlvfile.MlvReader(names[0]) as r:
        assert r.frame_count == len(frames)
        got = r.read_frames(0, len(frames), (W * H * 14 // 8 + 2 + 15) // 16 * 16)
        for k, f in enumerate(frames):
            assert bytes(got[k][:W * H * 14 // 8]) == synth.pack_bits(f).tobytes()[:W * H * 14 // 8]


def test_lzma_clip_with_hostile_size_fields_is_refused_not_allocated(reference, tmp_path):
    """The VIDF blockSize and the payload's 32-bit size word come from the file: buffers sized from them are bounded by what the
    chunk really holds and by the frame's packed size (the reference mallocs both as they stand, main.c:587-602)."""
    import resource
    import struct
    names, frames = lzma_clip(reference, tmp_path, n=3)
    raw = bytearray(open(names[0], "rb").read())
    stride = (W * H * 14 // 8 + 2 + 15) // 16 * 16
    last = raw.rfind(b"VIDF")
    # (1) the last frame's block claims 3.7 GiB
    bad = bytearray(raw)
    bad[last + 4:last + 8] = struct.pack("<I", 0xF0000000)
    open(names[0], "wb").write(bytes(bad))
    peak0 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
    with mlvfile.MlvReader(names[0]) as r:
        if r.frame_count == len(frames):                                  # (an index that drops the over-long block is fine as well)
            try:
                got = r.read_frames(len(frames) - 1, 1, stride)
            except lib.MlvfsAmdError as e:                                # refusing is fine too
                assert "mlv" in str(e)
            else:
                assert bytes(got[0][:W * H * 14 // 8]) == synth.pack_bits(frames[-1]).tobytes()[:W * H * 14 // 8]
    # (2) its payload's size word claims 4 GiB
    bad = bytearray(raw)
    vidf_hdr = 32                                                         # mlv_vidf_hdr_t (mlv.h:69-81)
    bad[last + vidf_hdr:last + vidf_hdr + 4] = struct.pack("<I", 0xFFFFFFF0)
    open(names[0], "wb").write(bytes(bad))
    with mlvfile.MlvReader(names[0]) as r:
        with pytest.raises(Exception, match="LZMA"):
            r.read_frames(r.frame_count - 1, 1, stride)
        assert bytes(r.read_frames(0, 1, stride)[0][:64]) == synth.pack_bits(frames[0]).tobytes()[:64]   # the others still read
    assert resource.getrusage(resource.RUSAGE_SELF).ru_maxrss - peak0 < 512 * 1024, "a buffer was sized from a hostile field (KiB)"


@pytest.mark.gpu
def test_lzma_clip_through_the_fused_pipeline(gpu, oracle, reference, tmp_path):
    from mlvfs_amd.stream import ClipStream
    w, h = 256, 130
    names, frames = lzma_clip(reference, tmp_path, n=5, w=w, h=h)
    s = ClipStream(w, h, 14, synth.BLACK, synth.WHITE)
    packed = s.upload_packed([synth.pack_bits(f) for f in frames])
    s.analyse_first_frame(packed, cs=5, bad_pix=1, stripes=True, rand_mode=1)
    out = np.zeros((len(frames), h, w), np.uint16)
    with mlvfile.MlvReader(names[0]) as r:
        r.process(s.clip, 0, len(frames), out, 5, True, True, batch=2)
    pixels = oracle.detect_bad_pixels(frames[0], synth.BLACK, 0)
    corr = None
    for k, f in enumerate(frames):
        img = oracle.chroma_smooth(oracle.apply_bad_pixels(f, synth.BLACK, pixels), synth.BLACK, 5)
        if corr is None:
            corr = oracle.stripes_compute(img, synth.BLACK, synth.WHITE, frame_size=w * h * 14 // 8)
        assert np.array_equal(out[k], oracle.stripes_apply(img, synth.BLACK, synth.WHITE, *corr)), k
    s.close()


# ------------------------------------------------------------------ GIF preview
def gif_of(amd, path):
    r = amd.mlvfs_amd_mlv_open(path.encode(), 0)
    assert r
    from mlvfs_amd import abi
    fh = abi.FrameHeaders()
    assert amd.mlvfs_amd_mlv_frame_headers(r, 0, C.byref(fh)) == 1
    n = amd.mlvfs_amd_gif_size(C.byref(fh))
    out = np.zeros(n, np.uint8)
    got = amd.mlvfs_amd_mlv_gif_data(r, lib.ptr(out), 0, n)
    # a window of the file, like a FUSE read
    part = np.zeros(1000, np.uint8)
    assert amd.mlvfs_amd_mlv_gif_data(r, lib.ptr(part), 777, 1000) == 1000 and bytes(part) == bytes(out[777:1777])
    amd.mlvfs_amd_mlv_close(r)
    assert got == n
    # gif.h's own two calls, exported by the library (main.c:1018-1022 calls them with the clip's path)
    assert amd.gif_get_size(C.byref(fh)) == n
    again = np.zeros(n, np.uint8)
    assert amd.gif_get_data(path.encode(), lib.ptr(again), 0, n) == n and bytes(again) == bytes(out)
    assert amd.gif_get_data(path.encode(), lib.ptr(part), 777, 1000) == 1000 and bytes(part) == bytes(out[777:1777])
    assert amd.gif_get_data(b"/nonexistent/clip.MLV", lib.ptr(part), 0, 1000) == 0
    return out.tobytes()


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["plain", "lzma", "lj92", "odd_width", "few_frames", "dark"])
def test_gif_preview_equals_the_reference(gpu, reference, tmp_path, kind):
    w, h, n, black = 256, 136, 23, synth.BLACK
    if kind == "odd_width":
        w, h = 250, 130                                                 # xRes / 4 * 4 != xRes: gif.c's row pitch is not the frame's
    if kind == "few_frames":
        n = 3                                                           # frames repeat: k * 3 / 10
    if kind == "dark":
        black = 8000
    frames = [synth.normal_frame(w, h, seed=6, frame=k, black=black) for k in range(n)]
    vc = 1
    if kind == "lzma":
        payloads, vc = [reference.lzma_payload(synth.pack_bits(f).tobytes()) for f in frames], 1 | 0x80
    elif kind == "lj92":
        from oracle import lj92_testenc
        payloads, vc = [], 1 | 0x100
        for f in frames:
            img = np.ascontiguousarray(np.block([[f[0::2, 0::2], f[0::2, 1::2]], [f[1::2, 0::2], f[1::2, 1::2]]]))
            st = lj92_testenc.encode(img, 6, 14)
            payloads.append((w * h * 2).to_bytes(4, "little") + bytes(st))
    else:
        payloads = [synth.pack_bits(f).tobytes() for f in frames]
    names = mlvfile.write_clip(str(tmp_path / "M02-0002.MLV"), payloads, w, h, black=black, video_class=vc)
    want = reference.gif(names[0])
    got = gif_of(gpu, names[0])
    assert len(got) == len(want) and got == want
    assert got[:6] == b"GIF89a" and got[-1] == 0x3B


@pytest.mark.gpu
def test_gif_preview_committed_hashes(gpu, tmp_path):
    """the same clips regenerated from the seeded generator, hashes of the reference's files committed (no reference build needed)"""
    import json
    gold = json.load(open(os.path.join(HERE, "golden", "gif_hashes.json")))
    for key, (w, h, n, black) in {"plain_256x136": (256, 136, 23, synth.BLACK), "odd_250x130": (250, 130, 7, synth.BLACK),
                                  "big_1920x1080": (1920, 1080, 11, synth.BLACK)}.items():
        frames = [synth.normal_frame(w, h, seed=6, frame=k, black=black) for k in range(n)]
        d = tmp_path / key
        d.mkdir()
        names = mlvfile.write_clip(str(d / "M03-0003.MLV"), [synth.pack_bits(f).tobytes() for f in frames], w, h, black=black)
        assert hashlib.sha256(gif_of(gpu, names[0])).hexdigest() == gold[key], key

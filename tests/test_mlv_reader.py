"""MLV container reader + prefetcher (csrc/mlvreader.cpp).

CPU (`-m "not gpu"`): index, .IDX file and frame count against the reference's own index.c (oracle/_ref) and against the
restatement oracle/mlv_container.py; per-frame headers against the restatement of main.c:429-558; payload reads.
GPU: file -> fused pipeline equals the oracle's process_frame on the same frames.
"""
import os
import shutil

import numpy as np
import pytest

from mlvfs_amd import mlvfile, synth
from oracle import mlv_container as orc

W, H = 64, 48


def payloads(n, w=W, h=H, seed=3):
    return [synth.pack_bits(synth.normal_frame(w, h, seed=seed, frame=k)).tobytes() for k in range(n)]


CLIPS = {
    "plain": dict(n=9),
    "plain_no_extras": dict(n=4, extras=False),
    "frame_space": dict(n=7, frame_space=96),
    "three_chunks": dict(n=11, chunks=3),
    "shuffled": dict(n=10, shuffle=True),
    "shuffled_chunks_space": dict(n=13, chunks=2, shuffle=True, frame_space=32),
}


@pytest.fixture(scope="module", params=sorted(CLIPS))
def clip(request, tmp_path_factory):
    kw = dict(CLIPS[request.param])
    n = kw.pop("n")
    d = tmp_path_factory.mktemp(request.param)
    pl = payloads(n)
    names = mlvfile.write_clip(str(d / "M27-1337.MLV"), pl, W, H, **kw)
    return names, pl, kw


def test_index_equals_reference_and_restatement(clip, reference):
    names, pl, _ = clip
    with mlvfile.MlvReader(names[0]) as r:
        got = r.xref()
        assert r.frame_count == len(pl) and r.chunk_count == len(names)
    assert got == orc.make_index(names)
    assert got == reference.mlv_index(names[0])                                    # index.c:216-341, 458-486
    assert not os.path.exists(names[0][:-3] + "IDX")


def test_index_equals_committed_reference_vectors(request, clip, tmp_path):
    """tests/golden/mlv_index.npz (made by the reference's index.c, tests/golden/make_mlv_golden.py)."""
    names, pl, _ = clip
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mlv_index.npz"))
    key = request.node.callspec.params["clip"]
    with mlvfile.MlvReader(names[0]) as r:
        assert r.xref() == gold[key + "_xref"].tobytes() == orc.make_index(names)
    assert orc.idx_file(names) == gold[key + "_idx"].tobytes()


def test_idx_file_equals_reference(clip, reference, tmp_path):
    names, pl, _ = clip
    a, b = tmp_path / "ours", tmp_path / "theirs"
    for d in (a, b):
        d.mkdir()
        for nm in names:
            shutil.copy(nm, d / os.path.basename(nm))
    base = os.path.basename(names[0])
    with mlvfile.MlvReader(str(a / base), use_idx_file=True) as r:                  # no .IDX yet: scan and save (index.c:426-456)
        ours = r.xref()
    assert reference.mlv_frame_count(str(b / base)) == len(pl)                      # the reference writes its own .IDX here
    idx_ours, idx_ref = (a / (base[:-3] + "IDX")).read_bytes(), (b / (base[:-3] + "IDX")).read_bytes()
    assert idx_ours == idx_ref == orc.idx_file([str(a / os.path.basename(nm)) for nm in names])
    assert reference.mlv_index(str(b / base), with_idx_file=True) == ours
    # an existing .IDX is believed (load_index, index.c:101-165), even when it lists fewer frames than the clip holds
    n_ent = (len(idx_ref) - 52 - 24) // 12
    cut = bytearray(idx_ref[: 52 + 24 + 12 * (n_ent - 1)])
    cut[52 + 4:52 + 8] = (24 + 12 * (n_ent - 1)).to_bytes(4, "little")
    cut[52 + 20:52 + 24] = (n_ent - 1).to_bytes(4, "little")
    (a / (base[:-3] + "IDX")).write_bytes(bytes(cut))
    (b / (base[:-3] + "IDX")).write_bytes(bytes(cut))
    with mlvfile.MlvReader(str(a / base), use_idx_file=True) as r:
        assert r.xref() == reference.mlv_index(str(b / base), with_idx_file=True) == bytes(cut[52:])


def test_frame_headers_equal_restatement(clip):
    names, pl, kw = clip
    with mlvfile.MlvReader(names[0]) as r:
        for k in list(range(len(pl))) + [len(pl), len(pl) + 5, -1]:
            ok, fh = r.frame_headers(k)
            want_ok, want = orc.frame_headers(names, k)
            assert ok == want_ok and bytes(fh) == want, k
        ok, fh = r.frame_headers(len(pl) - 1)
        assert ok == 1 and fh.rawi_hdr.xRes == W and fh.vidf_hdr.frameSpace == kw.get("frame_space", 0)
        if kw.get("extras", True) and not kw.get("shuffle"):
            assert fh.expo_hdr.isoValue == 800 and fh.lens_hdr.focalLength == 35          # the mid-clip EXPO / LENS blocks
            assert r.frame_headers(0)[1].expo_hdr.isoValue == 100
        if len(names) > 1:
            assert fh.file_hdr.fileNum == len(names) - 1                                    # the LAST chunk's MLVI wins (main.c:496-500)


def test_frame_headers_equal_the_reference_text(clip, reference):
    """mlv_get_frame_headers as main.c:429-558 defines it (the reference's own text, sliced into oracle/_ref by oracle/Makefile)
    against the reader of libmlvfs_amd.so and against the Python restatement oracle/mlv_container.py."""
    names, pl, kw = clip
    idx = names[0][:-3] + "IDX"
    if os.path.exists(idx):
        os.remove(idx)
    with mlvfile.MlvReader(names[0]) as r:
        for k in range(len(pl)):
            r.frame_headers(k)
        assert not os.path.exists(idx), "the library's reader (use_idx_file=0) wrote an index file"   # the reference's walk below does (index.c:458-470)
        for k in list(range(len(pl))) + [len(pl), len(pl) + 5]:
            ok, fh = r.frame_headers(k)
            want_ok, want = reference.mlv_frame_headers(names[0], k)
            rest_ok, rest = orc.frame_headers(names, k)
            assert ok == want_ok == rest_ok, k
            if want_ok:                          # (not found: the reference leaves part of the struct as the search left it)
                assert bytes(fh) == want == rest, k


def test_payloads(clip):
    names, pl, _ = clip
    stride = (len(pl[0]) + 2 + 15) // 16 * 16
    with mlvfile.MlvReader(names[0]) as r:
        got = r.read_frames(0, len(pl), stride, io_threads=3)
        for k, p in enumerate(pl):
            assert got[k, :len(p)].tobytes() == p and not got[k, len(p):].any(), k
        one = r.read_frames(len(pl) - 1, 1, stride)
        assert one[0, :len(pl[-1])].tobytes() == pl[-1]
        with pytest.raises(Exception):
            r.read_frames(len(pl) - 1, 2, stride)                                       # past the last frame
        with pytest.raises(Exception):
            r.read_frames(0, 1, len(pl[0]) - 16)                                        # stride smaller than a payload


def test_damaged_and_foreign_files(tmp_path, reference):
    pl = payloads(5)
    names = mlvfile.write_clip(str(tmp_path / "A.MLV"), pl, W, H, chunks=2)
    # truncated in the middle of the last frame: the index still lists the block, the payload read fails
    data = open(names[0], "rb").read()
    open(names[0], "wb").write(data[:-100])
    with mlvfile.MlvReader(names[0]) as r:
        assert r.xref() == reference.mlv_index(names[0]) == orc.make_index(names)
        stride = (len(pl[0]) + 2 + 15) // 16 * 16
        with pytest.raises(Exception):
            r.read_frames(0, r.frame_count, stride)
    # a second chunk from another recording (GUID mismatch) contributes nothing (index.c:271-279)
    other = mlvfile.write_clip(str(tmp_path / "B.MLV"), pl, W, H, chunks=2, guid=0x9999)
    shutil.copy(other[1], names[1])
    with mlvfile.MlvReader(names[0]) as r:
        assert r.xref() == reference.mlv_index(names[0]) == orc.make_index(names)
        assert r.frame_count == 3
    # garbage block size stops the scan of that chunk
    open(names[0], "ab").write(b"VIDF" + (5).to_bytes(4, "little") + bytes(8))
    with mlvfile.MlvReader(names[0]) as r:
        assert r.xref() == reference.mlv_index(names[0]) == orc.make_index(names)
    with pytest.raises(Exception):
        mlvfile.MlvReader(str(tmp_path / "missing.MLV"))


def test_compressed_clips_are_refused(tmp_path):
    names = mlvfile.write_clip(str(tmp_path / "C.MLV"), payloads(2), W, H, video_class=1 | 0x100)
    with mlvfile.MlvReader(names[0]) as r:
        assert r.frame_count == 2 and r.frame_headers(0)[0] == 1
        with pytest.raises(Exception, match="LJ92"):                      # (LZMA clips read like plain ones: tests/test_lzma_gif.py)
            r.read_frames(0, 1, 8192)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["packed14", "lj92"])
def test_file_to_gpu_pipeline_equals_oracle(gpu, oracle, tmp_path, kind):
    from mlvfs_amd.stream import ClipStream, to_numpy_u16
    w, h, n = 256, 130, 23
    frames = [synth.normal_frame(w, h, seed=5, frame=k) for k in range(n)]
    if kind == "packed14":
        pl = [synth.pack_bits(f).tobytes() for f in frames]
        names = mlvfile.write_clip(str(tmp_path / "G.MLV"), pl, w, h, chunks=2, frame_space=64, shuffle=True)
    else:                                                                      # main.c:617-681: size word + lossless JPEG of the tiled frame
        import struct
        from oracle import lj92_testenc as enc
        from test_lj92 import quadrants
        pl = [struct.pack("<I", w * h * 2) + enc.encode(quadrants(f), 6, 14) for f in frames]
        names = mlvfile.write_clip(str(tmp_path / "G.MLV"), pl, w, h, chunks=2, frame_space=64, shuffle=True, video_class=1 | 0x100)
    s = ClipStream(w, h, 14, synth.BLACK, synth.WHITE, device=0)
    packed0 = s.upload_packed([synth.pack_bits(frames[0])])
    s.analyse_first_frame(packed0, cs=5, bad_pix=1, stripes=True, rand_mode=1)
    pixels = oracle.detect_bad_pixels(frames[0], synth.BLACK, 0)
    want, corr = [], None
    for f in frames:
        img = oracle.chroma_smooth(oracle.apply_bad_pixels(f, synth.BLACK, pixels), synth.BLACK, 5)
        if corr is None:
            corr = oracle.stripes_compute(img, synth.BLACK, synth.WHITE, frame_size=w * h * 14 // 8)
        want.append(oracle.stripes_apply(img, synth.BLACK, synth.WHITE, *corr))
    with mlvfile.MlvReader(names[0]) as r:
        assert r.frame_count == n
        for batch, first, count in ((4, 0, n), (32, 3, 11), (1, n - 2, 2)):
            out = np.zeros((count, h, w), np.uint16)
            r.process(s.clip, first, count, out, cs=5, fix_pixels=True, stripes=True, batch=batch, io_threads=3)
            for k in range(count):
                assert np.array_equal(out[k], want[first + k]), (batch, first + k)
    s.close()


def test_reader_survives_mutated_files(tmp_path):
    """Random damage to block headers: opening, indexing, header gathering and payload reads answer or refuse."""
    pl = payloads(6)
    names = mlvfile.write_clip(str(tmp_path / "F.MLV"), pl, W, H, chunks=2, frame_space=16)
    good = [open(n, "rb").read() for n in names]
    rng = np.random.default_rng(3)
    stride = (len(pl[0]) + 2 + 15) // 16 * 16
    opened = 0
    for _ in range(150):
        for n, g in zip(names, good):
            b = bytearray(g)
            for _ in range(int(rng.integers(1, 8))):
                b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
            if rng.integers(0, 4) == 0:
                b = b[: int(rng.integers(16, len(b)))]
            open(n, "wb").write(bytes(b))
        try:
            with mlvfile.MlvReader(names[0]) as r:
                opened += 1
                assert r.xref() == orc.make_index(names)                          # whatever the damage, the same index
                for k in range(r.frame_count + 1):
                    r.frame_headers(k)
                try:
                    r.read_frames(0, r.frame_count, stride)
                except Exception:
                    pass
        except Exception:
            pass
    assert opened > 100


@pytest.mark.gpu
def test_virtual_dng_files_equal_the_reference(gpu, oracle, reference, tmp_path):
    """What MLVFS serves for <clip>/<frame>.dng is 65536 header bytes + the processed pixels (main.c:908-1005, 1444-1500).
    Here: container walk (reader) -> per-frame headers -> DNG header writer, and payloads -> fused GPU pipeline; against the
    reference's dng_get_header_data on the restated frame headers + the reference's own process_frame stages."""
    import ctypes as C
    from mlvfs_amd import lib
    from mlvfs_amd.stream import ClipStream
    w, h, n = 256, 130, 6
    frames = [synth.normal_frame(w, h, seed=8, frame=k) for k in range(n)]
    pl = [synth.pack_bits(f).tobytes() for f in frames]
    names = mlvfile.write_clip(str(tmp_path / "M11-0815.MLV"), pl, w, h, chunks=2, frame_space=32)
    L = lib.load()
    s = ClipStream(w, h, 14, synth.BLACK, synth.WHITE, device=0)
    s.analyse_first_frame(s.upload_packed([synth.pack_bits(frames[0])]), cs=5, bad_pix=1, stripes=True, rand_mode=1)
    corr = None
    with mlvfile.MlvReader(names[0]) as r:
        out = np.zeros((n, h, w), np.uint16)
        r.process(s.clip, 0, n, out, cs=5, fix_pixels=True, stripes=True, batch=4)
        for k in range(n):
            ok, fh = r.frame_headers(k)
            assert ok == 1
            hdr = np.zeros(65536, np.uint8)
            assert L.dng_get_header_data(C.byref(fh), lib.ptr(hdr), 0, 65536, 0.0, b"M11-0815") == 65536
            ours = hdr.tobytes() + out[k].tobytes()
            # the reference: headers as main.c:429-558 gathers them (restated), its header writer, its pixel stages
            want_ok, blob = orc.frame_headers(names, k)
            assert want_ok == 1
            _, ref_hdr, _ = reference.header_data(np.frombuffer(blob, np.uint8), 0, 65536, 0.0, b"M11-0815")
            packed = np.concatenate([np.frombuffer(pl[k], "<u2"), np.zeros(4, "<u2")])
            img, corr = reference.process_frame(packed, w, h, synth.BLACK, synth.WHITE, cs=5, bad_pix=1, stripes=1, correction=corr,
                                                guid=0x5EED0815)
            theirs = ref_hdr.tobytes() + img.tobytes()
            assert len(ours) == L.dng_get_size(C.byref(fh)) == len(theirs)
            assert ours == theirs, k
    s.close()


@pytest.mark.gpu
def test_process_refuses_a_clip_state_of_another_geometry(gpu, tmp_path):
    from mlvfs_amd.stream import ClipStream
    names = mlvfile.write_clip(str(tmp_path / "Q.MLV"), payloads(3), W, H)
    s = ClipStream(W + 16, H, 14, synth.BLACK, synth.WHITE, device=0)
    with mlvfile.MlvReader(names[0]) as r:
        with pytest.raises(Exception, match="clip state"):
            r.process(s.clip, 0, 3, np.zeros((3, H, W + 16), np.uint16), cs=0, fix_pixels=False, stripes=False)
    s.close()


@pytest.mark.gpu
@pytest.mark.parametrize("interp", [0, 1])
def test_dual_iso_clip_from_file_equals_the_checker(gpu, oracle, tmp_path, interp):
    """mlvfs_amd_mlv_process_dualiso: a two-chunk clip of dual-ISO frames with one normal frame in it, batches of 3 (so that the
    prefetch, the two frame slots and a short last batch are all exercised), against get_image_data + cr2hdr20_convert_data frame by
    frame in clip order (the table caches carry the first converted frame's white level through the clip)."""
    w, h, n = 416, 264, 8
    frames = [synth.normal_frame(w, h, seed=4) if k == 2 else synth.dual_iso_frame(w, h, seed=3 + k, frame=k) for k in range(n)]
    pl = [np.ascontiguousarray(synth.pack_bits(f), "<u2").tobytes() for f in frames]
    names = mlvfile.write_clip(str(tmp_path / "D.MLV"), pl, w, h, chunks=2, frame_space=16, shuffle=True)
    oracle.L.orc_dualiso_reset()
    gpu.mlvfs_amd_dualiso_reset()
    want = [oracle.cr2hdr20(f, synth.BLACK, synth.WHITE, interp, 1, 1, 0, reset=False) for f in frames]
    out = np.zeros((n, h, w), np.uint16)
    with mlvfile.MlvReader(names[0]) as r:
        res = r.process_dualiso(0, n, out, interp=interp, batch=3, io_threads=2)
    assert list(res) == [wr for wr, _, _ in want] == [1, 1, 0, 1, 1, 1, 1, 1]
    for k, (wr, img, _) in enumerate(want):
        assert np.array_equal(out[k], img if wr == 1 else frames[k]), k

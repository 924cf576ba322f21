"""F1 with the reference's REAL focus-pixel maps (mlvfs/data/*.fpm: four cameras x four raw geometries, up to 151 200 entries at
2592x1108), committed as coordinates (tests/golden/focus_maps.npz) with hashes of what the reference's own fix_focus_pixels makes
of seeded frames -- normal and dual-ISO rule, crop offsets 0 and non-zero (tests/golden/make_focus_golden.py).  cs.c:336-503."""
import ctypes as C
import os

import numpy as np
import pytest

import focus_maps
from mlvfs_amd import abi, lib, synth

BLACK, WHITE = synth.BLACK, synth.WHITE
G = focus_maps.golden()
CASES = G["cases"]
IDS = [f"{c['map']}-{c['w']}x{c['h']}-pan{c['pan'][0]},{c['pan'][1]}-{c['kind']}" for c in CASES]


def frame_for(case):
    f = synth.dual_iso_frame if case["kind"] == "dual_iso" else synth.normal_frame
    return f(case["w"], case["h"], seed=31)


def crop_of(case):
    return ((case["pan"][0] + 7) & ~7, case["pan"][1] & ~1)                 # cs.c:439-440


def test_committed_maps_are_the_reference_files():
    ref = "/root/reference/mlvfs/data"
    if not os.path.isdir(ref):
        pytest.skip("needs the reference tree")
    names = sorted(f[:-4] for f in os.listdir(ref) if f.endswith(".fpm"))
    assert len(names) == 16 and max(c["entries"] for c in CASES) == 151200
    for n in names:
        assert np.array_equal(focus_maps.load(n), np.loadtxt(os.path.join(ref, n + ".fpm"), dtype=np.int64).reshape(-1, 2)), n


@pytest.mark.parametrize("case", CASES, ids=IDS)
def test_oracle_reproduces_the_reference_on_real_maps(oracle, case):
    f = frame_for(case)
    out = oracle.apply_focus_pixels(f, BLACK, focus_maps.load(case["map"]), crop_of(case), case["dual_iso"])
    assert int((out != f).sum()) == case["changed"] and synth.fnv1a(out) == case["hash"]


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES, ids=IDS)
def test_drop_in_symbol_on_real_maps(gpu, tmp_path, monkeypatch, case):
    """fix_focus_pixels reads '<camera hex>_<raw w>x<raw h>.fpm' from the current directory like the reference; twice, the second
    time inside a frame bracket after an unpack (the recorded unpack has to run before the repair: the map is not a bad-pixel map)."""
    monkeypatch.chdir(tmp_path)
    focus_maps.write_fpm(tmp_path, case["map"])
    gpu.free_focus_pixel_maps()
    f = frame_for(case)
    fh = abi.make_frame_headers(case["w"], case["h"], black=BLACK, white=WHITE)
    fh.idnt_hdr.cameraModel = case["camera"]
    fh.rawi_hdr.raw_info.width, fh.rawi_hdr.raw_info.height = case["raw_w"], case["raw_h"]
    fh.vidf_hdr.panPosX, fh.vidf_hdr.panPosY = case["pan"]
    got = f.copy()
    gpu.fix_focus_pixels(C.byref(fh), lib.ptr(got), case["dual_iso"])
    assert int((got != f).sum()) == case["changed"] and synth.fnv1a(got) == case["hash"]
    got2 = np.zeros_like(f)
    packed = np.ascontiguousarray(synth.pack_bits(f), np.uint16)
    gpu.mlvfs_amd_frame_begin()
    assert gpu.dng_get_image_data(C.byref(fh), lib.ptr(packed), lib.ptr(got2), 0, got2.nbytes) == got2.nbytes
    gpu.fix_focus_pixels(C.byref(fh), lib.ptr(got2), case["dual_iso"])
    assert gpu.mlvfs_amd_frame_end() == 0
    assert synth.fnv1a(got2) == case["hash"]
    gpu.free_focus_pixel_maps()


@pytest.mark.gpu
@pytest.mark.parametrize("case", [c for c in CASES if c["kind"] == "normal" and c["w"] % 8 == 0], ids=[i for c, i in zip(CASES, IDS) if c["kind"] == "normal" and c["w"] % 8 == 0])
def test_fused_path_on_real_maps(gpu, oracle, case):
    """The batch path: the map set on the clip (kind 1 = focus pixels), unpack + repair in ONE fused pass, three frames per launch;
    then the same with cs5x5 behind the repair against the checker (the golden hash covers the repair alone)."""
    from mlvfs_amd.stream import ClipStream, to_numpy_u16
    w, h = case["w"], case["h"]
    s = ClipStream(w, h, 14, BLACK, WHITE, device=0, pan=tuple(case["pan"]))
    xy = focus_maps.load(case["map"])
    s.set_pixel_map(xy, kind=1)
    frames = [frame_for(case), synth.normal_frame(w, h, seed=32), frame_for(case)]
    packed = s.upload_packed([synth.pack_bits(f) for f in frames])
    got = to_numpy_u16(s.process(packed, cs=0, fix_pixels=True, stripes=False))
    assert synth.fnv1a(got[0]) == case["hash"] and np.array_equal(got[0], got[2])
    assert np.array_equal(got[1], oracle.apply_focus_pixels(frames[1], BLACK, xy, crop_of(case), 0))
    got5 = to_numpy_u16(s.process(packed[:1], cs=5, fix_pixels=True, stripes=False))
    assert np.array_equal(got5[0], oracle.chroma_smooth(got[0], BLACK, 5))
    s.close()

"""fix_focus_pixels through the drop-in symbol with the densest of the reference's real maps (151 200 entries, 2592x1108 raw): the
first call of a process (reads and parses '<camera>_<w>x<h>.fpm' from the current directory, builds the dependency levels) and the
calls after it, ms; the reference beside it when oracle/_ref is there."""
import ctypes as C, os, sys, tempfile, time
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import focus_maps
from mlvfs_amd import abi, lib, synth

L = lib.load(); assert L.mlvfs_amd_init(0) == 0
G = focus_maps.golden()
case = max((c for c in G["cases"] if c["kind"] == "normal"), key=lambda c: len(focus_maps.load(c["map"])))
print("map", case["map"], len(focus_maps.load(case["map"])), "entries; frame", case["w"], "x", case["h"])
tmp = tempfile.mkdtemp(); os.chdir(tmp)
focus_maps.write_fpm(tmp, case["map"])
f = synth.normal_frame(case["w"], case["h"], seed=31)
fh = abi.make_frame_headers(case["w"], case["h"], black=synth.BLACK, white=synth.WHITE)
fh.idnt_hdr.cameraModel = case["camera"]
fh.rawi_hdr.raw_info.width, fh.rawi_hdr.raw_info.height = case["raw_w"], case["raw_h"]
fh.vidf_hdr.panPosX, fh.vidf_hdr.panPosY = case["pan"]
t = []
for k in range(5):
    got = f.copy()
    t0 = time.perf_counter(); L.fix_focus_pixels(C.byref(fh), lib.ptr(got), 0); t.append((time.perf_counter() - t0) * 1e3)
print("fix_focus_pixels: " + " ".join(f"{x:.2f}" for x in t) + " ms (first call first); hash ok:", synth.fnv1a(got) == case["hash"])
try:
    from oracle import bindings
    if bindings.have_ref():
        Rf = bindings.Reference()
        for k in range(2):
            t0 = time.perf_counter(); Rf.fix_focus_pixels(f, synth.BLACK, 0, case["camera"], case["raw_w"], case["raw_h"], tuple(case["pan"])); t1 = time.perf_counter()
            print(f"reference call {k}: {(t1 - t0) * 1e3:.1f} ms")
except Exception as e:  # noqa: BLE001
    print("reference not timed:", e)

"""Time the full dual-ISO conversion (BASELINE.json config 4) on a device-resident 3584x1320 frame.
usage: python tools/dualiso_bench.py [interp_method] [reps]"""
import ctypes as C, sys, time, os
import numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from mlvfs_amd import lib, synth
import torch
interp = int(sys.argv[1]) if len(sys.argv) > 1 else 0
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
gpu = lib.load(); gpu.mlvfs_amd_init(0)
w, h = 3584, 1320
f = synth.dual_iso_frame(w, h)
src = torch.from_numpy(f.view(np.int16)).cuda()
geom = lib.Geom(w, h, 14, synth.BLACK, synth.WHITE, 0, 0)
devnull = os.open(os.devnull, os.O_WRONLY); saved = os.dup(1); os.dup2(devnull, 1)      # the reference's progress printf()s
try:
    for cs in (0, 5):
        ts = []
        for r in range(reps + 2):
            t = src.clone(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            rc = gpu.mlvfs_amd_cr2hdr20_dev(C.byref(geom), C.c_void_p(t.data_ptr()), interp, 1, 1, cs, None)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
            assert rc == 1
        ts = np.array(ts[2:]) * 1e3
        sys.stderr.write(f"dual-ISO {w}x{h} interp={interp} cs={cs}: {ts.mean():.2f} ms/frame (min {ts.min():.2f}) -> {1e3 / ts.mean():.1f} fps, {w * h / ts.mean() / 1e3:.1f} Mpix/s\n")
finally:
    os.dup2(saved, 1)

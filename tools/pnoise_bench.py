"""Time fix_pattern_noise (drop-in symbol, host buffers) on 3584x1320."""
import ctypes as C, sys, time, os
import numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from mlvfs_amd import lib, synth
gpu = lib.load(); gpu.mlvfs_amd_init(0)
w, h = 3584, 1320
f = synth.normal_frame(w, h, seed=1)
devnull = os.open(os.devnull, os.O_WRONLY); saved = os.dup(1); os.dup2(devnull, 1)
ts = []
for _ in range(7):
    img = f.copy()
    t0 = time.perf_counter()
    gpu.fix_pattern_noise(lib.ptr(img), w, h, synth.WHITE, 0)
    ts.append(time.perf_counter() - t0)
C.CDLL(None).fflush(None); os.dup2(saved, 1)
ts = np.array(ts[2:]) * 1e3
sys.stderr.write(f"fix_pattern_noise {w}x{h} (host buffers): {ts.mean():.2f} ms/frame (min {ts.min():.2f})\n")

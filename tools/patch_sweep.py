#!/usr/bin/env python3
"""cs5x5 + pixel map + stripes on the benchmark's frames with pixel maps of growing size (a focus-pixel map of an EOS M has
tens of thousands of entries): k_frame time (HIP events around the launch) and the whole process() call (k_pixfix + k_frame)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mlvfs_amd import lib
from mlvfs_amd.stream import ClipStream

W, H, F = 3584, 1320, 50
s = ClipStream(W, H)
L = s.L
base = s.synth_packed(8, seed=1)
packed = s.alloc_packed(F)
for i in range(0, F, 8):
    packed[i:i + 8] = base[:min(8, F - i)]
out = s.alloc_out(F)
s.set_stripes(1, [65536, 65536, 65354, 65738, 65241, 65868, 65450, 65640])
rng = np.random.default_rng(5)
for n in [int(x) for x in os.environ.get("PS_SIZES", "0,128,1000,4000,16000,64000").split(",")]:
    kind = os.environ.get("PS_KIND", "focus")
    if n:
        if kind == "focus":      # a regular grid like a focus-pixel map: every 8th column on every 12th row, until n entries
            ys, xs = np.mgrid[6:H - 6:12, 7:W - 8:8]
            idx = rng.permutation(ys.size)[:n]
            xy = np.stack([xs.reshape(-1)[idx], ys.reshape(-1)[idx]], 1)
        else:
            xy = np.stack([rng.integers(8, W - 8, n), rng.integers(8, H - 8, n)], 1)
        xy = np.unique(xy, axis=0)
        s.set_pixel_map(xy.astype(np.int32), kind=1 if kind == "focus" else 0)
    run = lambda: s.process(packed, out, cs=5, fix_pixels=bool(n), stripes=True)
    run(); torch.cuda.synchronize()
    ks, ws = [], []
    for _ in range(5):
        lib.check(L.mlvfs_amd_timer_begin(1))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(); e1.record(); torch.cuda.synchronize()
        ms = np.zeros(1, np.float32)
        L.mlvfs_amd_timer_end(lib.ptr(ms), 1)
        ks.append(float(ms[0])); ws.append(e0.elapsed_time(e1))
    print(f"{kind} map of {n:6d} entries: k_frame {np.median(ks) * 1e3 / F:6.2f} us/frame, whole call {np.median(ws) * 1e3 / F:6.2f} us/frame", flush=True)

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

# ---- the reference's REAL focus-pixel maps (mlvfs/data/*.fpm, committed as coordinates: tests/golden/focus_maps.npz) at the
# geometries they belong to: cs5x5 + the map + stripes, a batch of F frames.  The densest is 80000346 at 2592x1108: 151 200 entries.
if os.environ.get("PS_REAL", "1") == "1":
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    import focus_maps
    from mlvfs_amd import synth
    for name in (os.environ["PS_ONLY"],) if os.environ.get("PS_ONLY") else ("80000331_1808x727", "80000346_1808x727", "80000346_1872x1060", "80000346_2592x1108"):
        raw_w, raw_h = (int(v) for v in name.split("_")[1].split("x"))
        w, h = (raw_w - 80) // 16 * 16, (raw_h - 30) // 2 * 2
        xy = focus_maps.load(name)
        c = ClipStream(w, h)
        b8 = c.synth_packed(8, seed=1)
        pk = c.alloc_packed(F)
        for i in range(0, F, 8):
            pk[i:i + 8] = b8[:min(8, F - i)]
        o = c.alloc_out(F)
        c.set_stripes(1, [65536, 65536, 65354, 65738, 65241, 65868, 65450, 65640])
        res = {}
        for with_map in (False, True):
            if with_map:
                c.set_pixel_map(xy, kind=1)
            run = lambda: c.process(pk, o, cs=5, fix_pixels=with_map, stripes=True)
            run(); torch.cuda.synchronize()
            ws = []
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); run(); e1.record(); torch.cuda.synchronize()
                ws.append(e0.elapsed_time(e1))
            res[with_map] = np.median(ws) * 1e3 / F
        inside = int(((xy[:, 0] < w) & (xy[:, 1] < h)).sum())
        print(f"real map {name}: {len(xy):6d} entries ({inside} inside the {w}x{h} frame): whole call {res[True]:6.2f} us/frame with the map, "
              f"{res[False]:6.2f} without ({w * h / 1e6:.2f} Mpix)", flush=True)
        c.close()

#!/usr/bin/env python3
"""Kernel micro-benchmark: fused pipeline variants on a resident 3584x1320 stream
(interleaved rounds in one process, median of the HIP-event kernel timer)."""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mlvfs_amd import lib, synth
from mlvfs_amd.stream import ClipStream

W, H = (int(v) for v in os.environ.get("KB_SIZE", "3584x1320").split("x"))          # KB_SIZE=1736x976: a width that is no multiple of 16 (VEC = 0)
F = int(os.environ.get("KB_FRAMES", "50"))
s = ClipStream(W, H)
L = s.L
base = s.synth_packed(8, seed=1)
if os.environ.get("KB_KIND") == "colour_cast":          # footage-like colour balance with hard colour edges
    fr = [synth.colour_cast_frame(W, H, seed=11 + i) for i in range(8)]
    base = s.upload_packed([synth.pack_bits(f) for f in fr])
if os.environ.get("KB_KIND") == "low_light":            # underexposed footage: pixels at / below black, no colour edges
    fr = [synth.low_light_frame(W, H, seed=21 + i) for i in range(8)]
    print("low_light: %.2f %% of the pixels at or below black" % (100.0 * np.mean([(f <= synth.BLACK).mean() for f in fr])))
    base = s.upload_packed([synth.pack_bits(f) for f in fr])
packed = s.alloc_packed(F)
for i in range(0, F, 8):
    packed[i:i + 8] = base[:min(8, F - i)]
out = s.alloc_out(F)
if os.environ.get("KB_LAYOUT"):                         # 1 = plain, 2 = spread raw2ev table in LDS (default: decided from the first frame)
    s.set_t16_layout(int(os.environ["KB_LAYOUT"]))
frame0 = s.unpack(packed[:1])
s.detect_bad_pixels(frame0[0], 0)
s.set_stripes(1, [65536, 65536, 65354, 65738, 65241, 65868, 65450, 65640])
variants = {"m0": (0, True, True), "m2": (2, True, True), "m3": (3, True, True), "m5": (5, True, True),
            "m5_nopatch_nostripes": (5, False, False), "m5_patch_only": (5, True, False), "m5_stripes_only": (5, False, True), "unpack_only": (0, False, False),
            "m2_plain": (2, False, False), "m2_stripes": (2, False, True), "m3_plain": (3, False, False), "m3_stripes": (3, False, True)}       # (what k_frame_s takes: no pixel map)
if os.environ.get("KB_ONLY"):                           # KB_ONLY=m2,m5: just these rows (profiler runs)
    variants = {k: v for k, v in variants.items() if k in os.environ["KB_ONLY"].split(",")}
res = {k: [] for k in variants}
for rnd in range(int(os.environ.get("KB_ROUNDS", "7"))):
    for name, (cs, fp, st) in variants.items():
        lib.check(L.mlvfs_amd_timer_begin(1))
        s.process(packed, out, cs=cs, fix_pixels=fp, stripes=st)
        torch.cuda.synchronize()
        ms = np.zeros(1, np.float32)
        n = L.mlvfs_amd_timer_end(lib.ptr(ms), 1)
        if n == 0:   # unpack-only path is not k_frame: time with torch events
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); s.process(packed, out, cs=cs, fix_pixels=fp, stripes=st); e1.record(); torch.cuda.synchronize()
            ms[0] = e0.elapsed_time(e1)
        res[name].append(float(ms[0]))
npx = W * H * F
for k, v in res.items():
    m = float(np.median(v[1:]))
    print(f"{k:24s} {m*1e3/F:8.2f} us/frame  {F/m*1e3:10.0f} fps  {npx*3.75/m/1e6:8.1f} GB/s  ({100*npx*3.75/m/1e6/8000:.1f}% of 8 TB/s)")

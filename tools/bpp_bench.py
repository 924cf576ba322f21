"""tools/bpp_bench.py -- the fused pipeline on 12- and 10-bit clips (ML's reduced bit depths) against 14-bit, 3584x1320, frames in HBM:
microseconds per frame of process() for cs0 / cs2x2 / cs5x5 (+ pixel map + stripes)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mlvfs_amd import lib, synth
from mlvfs_amd.stream import ClipStream

W, H = (int(v) for v in os.environ.get("KB_SIZE", "3584x1320").split("x"))
F = int(os.environ.get("KB_FRAMES", "48"))
for bpp in (14, 12, 10):
    black, white = synth.BLACK >> (14 - bpp), synth.WHITE >> (14 - bpp)
    s = ClipStream(W, H, bpp, black, white)
    fr = [(synth.normal_frame(W, H, seed=1, frame=k) >> (14 - bpp)).astype(np.uint16) for k in range(4)]
    base = s.upload_packed([synth.pack_bits(f, bpp) for f in fr])
    packed = s.alloc_packed(F)
    for i in range(0, F, 4): packed[i:i + 4] = base[:min(4, F - i)]
    out = s.alloc_out(F)
    frame0 = s.unpack(packed[:1])
    s.detect_bad_pixels(frame0[0], 0)
    s.set_stripes(1, [65536, 65536, 65354, 65738, 65241, 65868, 65450, 65640])
    line = f"{bpp:2d} bit:"
    for cs in (0, 2, 5):
        for _ in range(2): s.process(packed, out, cs=cs, fix_pixels=True, stripes=True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): s.process(packed, out, cs=cs, fix_pixels=True, stripes=True)
        torch.cuda.synchronize()
        line += f"  cs{cs}: {(time.perf_counter() - t0) / 5 / F * 1e6:6.2f} us/frame"
    print(line, flush=True)
    s.close()

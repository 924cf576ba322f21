"""tools/secondary_stats.py -- the per-clip / per-frame side paths at 3584x1320 on frames resident in HBM, a few calls each, for
`rocprofv3 --kernel-trace --stats`: deflicker, bad-pixel detection, the first frame's stripes analysis, unpack, the separate stages."""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mlvfs_amd import lib, synth
from mlvfs_amd.stream import ClipStream

w, h = 3584, 1320
s = ClipStream(w, h, 14, synth.BLACK, synth.WHITE, device=0)
L = s.L
packed = s.synth_packed(4, seed=1)
def timed(name, fn, n=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    sys.stderr.write(f"{name:28s} {(time.perf_counter() - t0) / n * 1e3:8.3f} ms per call\n")
frames = s.unpack(packed)
timed("unpack x4", lambda: s.unpack(packed))
geom = lib.Geom(w, h, 14, synth.BLACK, synth.WHITE, 0, 0)
eb = np.zeros(2, np.int32)
timed("deflicker", lambda: lib.check(L.mlvfs_amd_deflicker_dev(C.byref(geom), C.c_void_p(frames[0].data_ptr()), w * h * 2, 3072, lib.ptr(eb), None)))
timed("detect_bad_pixels", lambda: s.detect_bad_pixels(frames[0], 0))
timed("detect_bad_pixels aggressive", lambda: s.detect_bad_pixels(frames[0], 1))
timed("stripes_compute", lambda: s.stripes_compute(frames[0]))
f2 = frames.clone()
timed("fix_pixels x4", lambda: s.fix_pixels(f2))
timed("stripes_apply x4", lambda: s.stripes_apply(f2))
for m in (2, 3, 5):
    timed(f"chroma_smooth {m}x{m} x4", lambda: s.chroma_smooth(f2, m))
s.close()

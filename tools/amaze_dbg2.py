import ctypes as C, sys
import numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from mlvfs_amd import lib, synth
from oracle.bindings import Oracle
import torch
gpu = lib.load(); gpu.mlvfs_amd_init(0)
o = Oracle()
def run(raw):
    h, w = raw.shape
    d_raw = torch.from_numpy(raw).cuda()
    out = [torch.full((h, w), float("nan"), dtype=torch.float32, device="cuda") for _ in range(3)]
    rc = gpu.mlvfs_amd_amaze_demosaic_dev(C.c_void_p(d_raw.data_ptr()), w, h, *[C.c_void_p(t.data_ptr()) for t in out], None)
    torch.cuda.synchronize()
    return [t.cpu().numpy() for t in out]
for (w, h) in [(260, 150)]:
    raw = synth.amaze_plane(w, h, w * 7 + h)
    raw[::2, ::2] *= 1.0 + 0.5 * ((np.arange(w)[None, ::2] // 3) % 2)
    raw = raw.clip(0, 0xFFFFF).astype(np.float32)
    got = run(raw); want = o.amaze_demosaic(raw)
    for n, g, x in zip("rgb", got, want):
        bad = g.view(np.uint32) != x.view(np.uint32)
        ys, xs = np.nonzero(bad)
        print(w, h, n, bad.sum(), "nan", np.isnan(g).sum(), "maxabs", np.nanmax(np.abs(g - x)))
        print("  cols", np.unique(xs)[:30], "rows", np.unique(ys)[:20], "...")
        print("  ", [(int(y), int(xx), float(g[y, xx]), float(x[y, xx])) for y, xx in list(zip(ys, xs))[:5]])

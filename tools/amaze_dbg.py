"""debug: GPU AMaZE planes vs oracle over a sweep of sizes with strong textures"""
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
    assert rc == 0
    torch.cuda.synchronize()
    return [t.cpu().numpy() for t in out]
rng = np.random.default_rng(5)
sizes = [(int(rng.integers(9, 150)) * 4, int(rng.integers(37, 500))) for _ in range(400)]
sizes += [(w, h) for w in range(264, 424, 4) for h in (150, 264)] + [(388, h) for h in range(132, 300, 4)]
nbad = 0
for (w, h) in sizes:
    raw = synth.amaze_plane(w, h, w * 7 + h)
    raw[::2, ::2] *= 1.0 + 0.5 * ((np.arange(w)[None, ::2] // 3) % 2)
    raw = raw.clip(0, 0xFFFFF).astype(np.float32)
    got = run(raw); got = run(raw) if (w + h) % 3 == 0 else got; want = o.amaze_demosaic(raw)
    bad = [int((g.view(np.uint32) != x.view(np.uint32)).sum()) for g, x in zip(got, want)]
    if any(bad):
        nbad += 1
        k = int(np.argmax(bad))
        g, x = got[k], want[k]
        ys, xs = np.nonzero(g.view(np.uint32) != x.view(np.uint32))
        print(w, h, "BAD", bad, "rows", (ys.min(), ys.max()) if len(ys) else None, "cols", (xs.min(), xs.max()) if len(xs) else None, flush=True)
print("sizes", len(sizes), "bad", nbad)

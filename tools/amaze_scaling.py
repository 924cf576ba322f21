"""How k_amaze's time grows with the number of tiles in one launch: flat up to the number of tiles the chip holds at once, then
stepwise.  Planes of full tiles only (width and height = 128 n - 16 + 144 ...): mlvfs_amd_amaze_demosaic_dev."""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mlvfs_amd import lib, synth
import torch
L = lib.load(); L.mlvfs_amd_init(0)
for tx, ty in ((4, 4), (8, 4), (8, 8), (16, 8), (16, 12), (16, 16), (32, 12), (32, 16), (32, 24), (32, 32)):
    w, h = 128 * tx + 16, 128 * ty + 16              # tiles_x = (w + 16 + 127) / 128 = tx + 1 with a narrow last column ... keep it simple: report the launch's tile count
    raw = torch.from_numpy(synth.amaze_plane(w, h, 1)).cuda()
    out = [torch.empty((h, w), dtype=torch.float32, device="cuda") for _ in range(3)]
    def run():
        rc = L.mlvfs_amd_amaze_demosaic_dev(C.c_void_p(raw.data_ptr()), w, h, *[C.c_void_p(o.data_ptr()) for o in out], None)
        assert rc == 0
    run(); torch.cuda.synchronize()
    ts = []
    for _ in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    tiles = ((w + 16 + 127) // 128) * ((h + 16 + 127) // 128)
    print(f"{w:5d} x {h:5d}: {tiles:5d} tiles  {min(ts):7.3f} ms  -> {min(ts) / tiles * 1e3:7.2f} us per tile, {min(ts) / max(1, -(-tiles // 256)) :6.3f} ms per round of 256", flush=True)

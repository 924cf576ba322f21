"""Time of k_amaze_rows alone (MLVFS_AMD_AMAZE_ROWS_ONLY=1: the incomplete tiles' launches are left out) on a plane of 48 x 45 = 2160
complete tiles, the tile count of a batch of 8 frames of 3584x1320; MLVFS_AMD_AMAZE_ROWS_SKIP switches passes off (timing only)."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MLVFS_AMD_AMAZE_ROWS_ONLY", "1")
os.environ.setdefault("MLVFS_AMD_AMAZE_ROWS", "1")
from mlvfs_amd import lib, synth
import torch
L = lib.load(); L.mlvfs_amd_init(0)
tx, ty = (int(v) for v in os.environ.get("TILES", "48x45").split("x"))
w, h = 128 * tx + 32 + 128, 128 * ty + 32 + 128
raw_np = synth.amaze_plane(w, h, 1)
raw_np[::2, ::2] *= 1.0 + 0.5 * ((np.arange(w)[None, ::2] // 3) % 2)
raw = torch.from_numpy(raw_np.clip(0, 0xFFFFF).astype(np.float32)).cuda()
out = [torch.empty((h, w), dtype=torch.float32, device="cuda") for _ in range(3)]
def run():
    rc = L.mlvfs_amd_amaze_demosaic_dev(C.c_void_p(raw.data_ptr()), w, h, *[C.c_void_p(o.data_ptr()) for o in out], None)
    assert rc == 0
run(); torch.cuda.synchronize()
ts = []
for _ in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
nfx, nfy = (w - 144) // 128 + 1, (h - 144) // 128 + 1
print(f"skip={os.environ.get('MLVFS_AMD_AMAZE_ROWS_SKIP', '0'):>6s}: {w}x{h}, ~{nfx * nfy} complete tiles: {min(ts):7.3f} ms (median {sorted(ts)[2]:.3f})", flush=True)

"""lj92_encode through the drop-in symbol: 3584x1320, host memory in and out; the reference's encoder beside it when oracle/_ref is there."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from mlvfs_amd import lib, lj92

L = lib.load()
assert L.mlvfs_amd_init(0) == 0
rng = np.random.default_rng(1)
w, h = 3584, 1320
img = np.clip(rng.normal(3000, 40, (h, w)) + np.linspace(0, 6000, w)[None, :], 0, 16383).astype(np.uint16)
s = lj92.encode(img, w, h, 14)
t = []
for _ in range(10):
    t0 = time.perf_counter(); s = lj92.encode(img, w, h, 14); t.append(time.perf_counter() - t0)
print(f"lj92_encode {w}x{h}: {min(t) * 1e3:.2f} ms best, {np.median(t) * 1e3:.2f} ms median, {len(s)} bytes ({len(s) * 8 / (w * h):.2f} bits/px)")
try:
    from oracle import bindings
    if bindings.have_ref():
        R = bindings.Reference()
        t0 = time.perf_counter(); r = R.lj92_encode_tile(img, w, h, 14); t1 = time.perf_counter()
        print(f"reference encoder, one host core: {(t1 - t0) * 1e3:.1f} ms, identical: {r == s}")
except Exception as e:  # noqa: BLE001
    print("reference not available:", e)

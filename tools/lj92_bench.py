#!/usr/bin/env python3
"""LJ92 decode throughput: N 3584x1320 14-bit frames compressed by the reference's encoder (oracle/_ref, predictor 6),
decoded by mlvfs_amd_lj92_decode_dev from host memory into HBM (H2D of the compressed bytes included), against the
reference's decoder on one host core.   usage: python tools/lj92_bench.py [frames]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mlvfs_amd import lib, lj92, synth
from oracle.bindings import Reference

W, H = 3584, 1320
N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
ref = Reference()
def quadrants(f):
    return np.ascontiguousarray(np.block([[f[0::2, 0::2], f[0::2, 1::2]], [f[1::2, 0::2], f[1::2, 1::2]]]))
frames = [synth.normal_frame(W, H, seed=1, frame=k) for k in range(4)]
t0 = time.perf_counter(); streams4 = [ref.lj92_encode(quadrants(f), 14) for f in frames]; t_enc = (time.perf_counter() - t0) / 4
t0 = time.perf_counter(); st, img = ref.lj92_decode(streams4[0]); t_dec = time.perf_counter() - t0
assert st == 0
print(f"compressed {len(streams4[0]) / 1e6:.2f} MB per frame ({len(streams4[0]) * 8 / W / H:.2f} bits/px); reference on one core: encode {t_enc * 1e3:.0f} ms, decode {t_dec * 1e3:.0f} ms per frame", flush=True)
lib.load().mlvfs_amd_init(0)
streams = [streams4[k % 4] for k in range(N)]
out = torch.empty((N, H, W), dtype=torch.int16, device="cuda")
pinned = [torch.from_numpy(np.frombuffer(s, np.uint8).copy()).pin_memory() for s in streams4]     # what the reader's staging is
for kind, src in (("pageable", streams), ("page-locked", [pinned[k % 4].numpy() for k in range(N)])):
  print("compressed bytes in", kind, "host memory")
  for batch in ([int(b) for b in os.environ['LJ_BATCHES'].split(',')] if os.environ.get('LJ_BATCHES') else (1, 4, 16, N)):
    lj92.decode_frames(src[:batch], W, H, out=out[:batch]); torch.cuda.synchronize()
    reps = max(1, 64 // batch)
    t0 = time.perf_counter()
    for _ in range(reps):
        lj92.decode_frames(src[:batch], W, H, out=out[:batch])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"  GPU decode, {batch:3d} frames per call: {dt / batch * 1e3:7.3f} ms per frame  {batch / dt:8.0f} fps  {batch * W * H / dt / 1e9:6.2f} Gpix/s", flush=True)
got = out.cpu().numpy().view(np.uint16)
assert all(np.array_equal(got[k], frames[k % 4]) for k in range(N)), "decoded frames differ from the originals"
print("all frames identical to the originals")
# the kernels alone: the compressed bytes of the call before are still in the arena (MLVFS_AMD_LJ92_NOUPLOAD=1, a measurement switch of
# csrc/lj92.cpp: a call with the same frames skips their upload) -- 4.5 MB per frame over a 50 GB/s link is 0.09 ms by itself
if os.environ.get("MLVFS_AMD_LJ92_NOUPLOAD") == "1":
    print("(the rows above, second call on: no upload -- kernels only)")

#!/usr/bin/env python3
"""File-to-frames throughput of a DUAL-ISO clip (BASELINE.json config 4, end to end): a synthetic 3584x1320 14-bit .MLV of dual-ISO
frames on the local disk (page cache warm) -> mlvfs_amd_mlv_process_dualiso (reader threads prefetch batch k+1, batch k is unpacked
and converted in one submission, batch k-1 travels back) -> 16-bit frames in host memory.  PCIe- and file-read-inclusive.
usage: python tools/mlv_e2e_dualiso_bench.py [frames] [dir]"""
import os, sys, time, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mlvfs_amd import lib, mlvfile, synth

W, H = 3584, 1320
N = int(sys.argv[1]) if len(sys.argv) > 1 else 96
d = sys.argv[2] if len(sys.argv) > 2 else tempfile.mkdtemp(prefix="mlvbench")
L = lib.load(); L.mlvfs_amd_init(0)
frames = [synth.dual_iso_frame(W, H, seed=3, frame=k) for k in range(4)]
pl = [np.ascontiguousarray(synth.pack14(f).astype("<u2")).tobytes() + b"\0" * 4 for f in frames]
names = mlvfile.write_clip(os.path.join(d, "DUAL.MLV"), [pl[k % 4] for k in range(N)], W, H, chunks=2, extras=True)
r = mlvfile.MlvReader(names[0])
out = torch.empty((N, H, W), dtype=torch.int16, pin_memory=True).numpy().view(np.uint16)
devnull = os.open(os.devnull, os.O_WRONLY); saved = os.dup(1); os.dup2(devnull, 1)      # the reference's progress printf()s
lines = []
try:
    for batch, io in ((4, 4), (8, 8), (16, 8)):
        res = r.process_dualiso(0, N, out, interp=0, batch=batch, io_threads=io)
        t0 = time.perf_counter()
        res = r.process_dualiso(0, N, out, interp=0, batch=batch, io_threads=io)
        dt = time.perf_counter() - t0
        assert int(res.sum()) == N and np.array_equal(out[0], out[4]) and not np.array_equal(out[0], frames[0])
        lines.append(f"file -> GPU -> host  batch={batch:3d} io_threads={io:2d}  unpack + cr2hdr20 (amaze-edge, fullres, alias map) {N / dt:7.1f} fps  {N * W * H / dt / 1e6:8.0f} Mpix/s")
finally:
    os.dup2(saved, 1)
print("\n".join(lines), flush=True)
r.close()
for n in names: os.remove(n)

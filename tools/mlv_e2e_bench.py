#!/usr/bin/env python3
"""File-to-frames throughput: a synthetic 3584x1320 14-bit .MLV on the local disk (page cache warm after writing) ->
mlvfs_amd_mlv_process (reader threads prefetch batch k+1 into page-locked staging while batch k is on the GPU) ->
16-bit frames in host memory.  PCIe- and file-read-inclusive; never bench.py's `value`.
usage: python tools/mlv_e2e_bench.py [frames] [dir]        MLV_E2E_LJ92=1: the clip's payloads are LJ92-compressed
(reference encoder, oracle/_ref), so the path is file -> LJ92 decode on the GPU -> stages -> host."""
import os, sys, time, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mlvfs_amd import mlvfile, synth
from mlvfs_amd.stream import ClipStream

W, H = 3584, 1320
N = int(sys.argv[1]) if len(sys.argv) > 1 else 192
d = sys.argv[2] if len(sys.argv) > 2 else tempfile.mkdtemp(prefix="mlvbench")
s = ClipStream(W, H)
base = s.synth_packed(8, seed=1)
s.analyse_first_frame(base, cs=5, bad_pix=1, stripes=True, rand_mode=1)
nbytes = W * H * 14 // 8
pl = [bytes(base[k % 8].cpu().numpy()[:nbytes]) for k in range(8)]
LJ = os.environ.get("MLV_E2E_LJ92") == "1"
if LJ:
    import struct
    from oracle.bindings import Reference          # only to MAKE the compressed test clip
    ref = Reference()
    frames8 = s.unpack(base).cpu().numpy().view(np.uint16)
    quad = lambda f: np.ascontiguousarray(np.block([[f[0::2, 0::2], f[0::2, 1::2]], [f[1::2, 0::2], f[1::2, 1::2]]]))
    pl = [struct.pack("<I", W * H * 2) + ref.lj92_encode(quad(frames8[k]), 14) for k in range(8)]
t0 = time.perf_counter()
names = mlvfile.write_clip(os.path.join(d, "BENCH.MLV"), [pl[k % 8] for k in range(N)], W, H, chunks=2, extras=True, video_class=1 | (0x100 if LJ else 0))
print(f"wrote {N} frames ({sum(os.path.getsize(n) for n in names) / 1e9:.2f} GB, {len(names)} chunks) in {time.perf_counter() - t0:.1f} s", flush=True)
t0 = time.perf_counter()
r = mlvfile.MlvReader(names[0])
print(f"open + index + per-frame headers of {r.frame_count} frames: {(time.perf_counter() - t0) * 1e3:.1f} ms", flush=True)
out = torch.empty((N, H, W), dtype=torch.int16, pin_memory=True).numpy().view(np.uint16)
want = None
for batch, io in ((16, 4), (32, 8), (32, 16), (64, 16)):
    r.process(s.clip, 0, N, out, cs=5, fix_pixels=True, stripes=True, batch=batch, io_threads=io)
    t0 = time.perf_counter()
    r.process(s.clip, 0, N, out, cs=5, fix_pixels=True, stripes=True, batch=batch, io_threads=io)
    dt = time.perf_counter() - t0
    if want is None: want = out[:8].copy()
    assert np.array_equal(out[8:16], want), "frames repeat every 8: outputs must too"
    print(f"file -> GPU -> host  batch={batch:3d} io_threads={io:2d}  unpack+badpix+cs5x5+stripes {N / dt:8.0f} fps  {N * W * H / dt / 1e6:9.0f} Mpix/s", flush=True)
r.close(); s.close()
for n in names: os.remove(n)

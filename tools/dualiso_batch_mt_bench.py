"""Batched dual-ISO conversions from T host threads at once (each its own HIP stream and batch of N frames in HBM): what one
process reaches when the analysis kernels of one batch overlap the AMaZE tiles of another.
usage: python tools/dualiso_batch_mt_bench.py [threads] [batch] [batches per thread]"""
import ctypes as C, json, os, sys, threading, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mlvfs_amd import lib, synth
import torch

T = int(sys.argv[1]) if len(sys.argv) > 1 else 2
N = int(sys.argv[2]) if len(sys.argv) > 2 else 8
REPS = int(sys.argv[3]) if len(sys.argv) > 3 else 4
L = lib.load(); L.mlvfs_amd_init(0)
w, h = 3584, 1320
f = synth.dual_iso_frame(w, h, seed=3)
src = torch.from_numpy(f.view(np.int16)).cuda()
geom = lib.Geom(w, h, 14, synth.BLACK, synth.WHITE, 0, 0)
devnull = os.open(os.devnull, os.O_WRONLY); saved = os.dup(1); os.dup2(devnull, 1)
bufs = [src.unsqueeze(0).repeat(N, 1, 1).contiguous() for _ in range(T)]
streams = [torch.cuda.Stream() for _ in range(T)]
torch.cuda.synchronize()
t_begin, t_end, oks = [0.0] * T, [0.0] * T, [0] * T
go = threading.Barrier(T)

def worker(i):
    L.mlvfs_amd_init(0)
    res = np.zeros(N, np.int32)
    sp = C.c_void_p(streams[i].cuda_stream)
    def run():
        with torch.cuda.stream(streams[i]):                 # fresh frames (a device copy on the same stream: 0.15 ms per 8 frames)
            bufs[i].copy_(src.unsqueeze(0).expand(N, -1, -1))
        rc = L.mlvfs_amd_cr2hdr20_batch_dev(C.byref(geom), C.c_void_p(bufs[i].data_ptr()), w * h * 2, N, 0, 1, 1, 0, lib.ptr(res), sp)
        streams[i].synchronize()
        return rc == 0 and int(res.sum()) == N
    run()                                      # the thread's work buffers
    go.wait()
    t_begin[i] = time.perf_counter()
    ok = 0
    for r in range(REPS):
        ok += run()
    t_end[i] = time.perf_counter()
    oks[i] = ok

try:
    th = [threading.Thread(target=worker, args=(i,)) for i in range(T)]
    for t in th: t.start()
    for t in th: t.join()
finally:
    os.dup2(saved, 1)
dt = max(t_end) - min(t_begin)
print(json.dumps({"threads": T, "batch": N, "batches_per_thread": REPS, "ok": oks, "conversions_per_s": round(T * N * REPS / dt, 1)}))

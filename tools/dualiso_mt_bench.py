"""Throughput of the full dual-ISO conversion with several frames in flight (one host thread + HIP stream per frame):
the single-frame latency (tools/dualiso_bench.py) is dominated by host decisions between kernels and by the 300-workgroup
AMaZE launch on 256 CUs, both of which overlap across frames.
usage: python tools/dualiso_mt_bench.py [interp_method] [frames_per_thread]"""
import ctypes as C, os, sys, threading, time
import numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from mlvfs_amd import lib, synth
import torch
interp = int(sys.argv[1]) if len(sys.argv) > 1 else 0
per_thread = int(sys.argv[2]) if len(sys.argv) > 2 else 12
gpu = lib.load(); gpu.mlvfs_amd_init(0)
w, h = 3584, 1320
f = synth.dual_iso_frame(w, h)
src = torch.from_numpy(f.view(np.int16)).cuda()
geom = lib.Geom(w, h, 14, synth.BLACK, synth.WHITE, 0, 0)
devnull = os.open(os.devnull, os.O_WRONLY); saved = os.dup(1); os.dup2(devnull, 1)      # the reference's progress printf()s


def worker(n, bufs, warm, stream, out, start, t_begin, t_end, idx):
    # steady state: one conversion of warm-up per thread (its stream and work buffers), clock (from the first worker that is ready to the last that is done); the thread's end (buffers
    # released, a device-wide synchronisation) is outside the clock
    gpu.mlvfs_amd_cr2hdr20_dev(C.byref(geom), C.c_void_p(warm.data_ptr()), interp, 1, 1, 0, C.c_void_p(stream.cuda_stream))
    stream.synchronize()
    t_begin[idx] = time.perf_counter()                 # (no barrier: the workers of a pool are not in step)
    ok = 0
    for k in range(n):
        ok += gpu.mlvfs_amd_cr2hdr20_dev(C.byref(geom), C.c_void_p(bufs[k].data_ptr()), interp, 1, 1, 0, C.c_void_p(stream.cuda_stream))
    stream.synchronize()
    t_end[idx] = time.perf_counter()
    out.append(ok)


try:
    ref = None
    for threads in [int(x) for x in os.environ.get("DI_THREADS", "1,2,4,8,16").split(",")]:
        streams = [torch.cuda.Stream() for _ in range(threads)]
        bufs = [[src.clone() for _ in range(per_thread)] for _ in range(threads)]
        warm = [src.clone() for _ in range(threads)]
        torch.cuda.synchronize()
        res, start, t_begin, t_end = [], None, [0.0] * threads, [0.0] * threads
        ths = [threading.Thread(target=worker, args=(per_thread, bufs[i], warm[i], streams[i], res, start, t_begin, t_end, i)) for i in range(threads)]
        for t in ths: t.start()
        for t in ths: t.join()
        torch.cuda.synchronize()
        dt = max(t_end) - min(t_begin)
        assert sum(res) == threads * per_thread
        if ref is None: ref = bufs[0][0].clone()
        assert all(torch.equal(b, ref) for bb in bufs for b in bb), "results differ between threads"
        n = threads * per_thread
        sys.stderr.write(f"dual-ISO {w}x{h} interp={interp}: {threads:2d} frames in flight: {n / dt:7.1f} conversions/s, {dt / n * 1e3:.2f} ms per frame, {w * h * n / dt / 1e9:.2f} Gpix/s\n")
finally:
    os.dup2(saved, 1)

"""Throughput of the batched dual-ISO conversion (BASELINE.json config 4: 3584x1320, --amaze-edge, full-res, alias map) on frames
resident in HBM: mlvfs_amd_cr2hdr20_batch_dev with N frames per submission, one host thread, one stream.
usage: python tools/dualiso_batch_bench.py [batch sizes, comma separated] [batches per size] [interp]"""
import ctypes as C, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mlvfs_amd import lib, synth
import torch

sizes = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "1,2,4,8,16").split(",")]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
interp = int(sys.argv[3]) if len(sys.argv) > 3 else 0
cs = int(os.environ.get('DI_BENCH_CS', '0'))                  # chroma smoothing of the conversion: 0, 2, 3, 5
L = lib.load(); L.mlvfs_amd_init(0)
w, h = (int(v) for v in os.environ.get('DI_BENCH_SIZE', '3584x1320').split('x'))
frames = [synth.dual_iso_frame(w, h, seed=3, frame=k) for k in range(2)]
src = [torch.from_numpy(f.view(np.int16)).cuda() for f in frames]
geom = lib.Geom(w, h, 14, synth.BLACK, synth.WHITE, 0, 0)
own = torch.cuda.Stream() if os.environ.get('DI_BENCH_STREAM') == '1' else None        # the caller's own stream instead of the library's
sp = C.c_void_p(own.cuda_stream) if own else None
devnull = os.open(os.devnull, os.O_WRONLY); saved = os.dup(1); os.dup2(devnull, 1)      # the reference's progress printf()s
out = {}
try:
    for n in sizes:
        buf = torch.empty((n, h, w), dtype=torch.int16, device="cuda")
        res = np.zeros(n, np.int32)
        def fill():
            for k in range(n):
                buf[k] = src[k & 1]
            torch.cuda.synchronize()
        def run():
            rc = L.mlvfs_amd_cr2hdr20_batch_dev(C.byref(geom), C.c_void_p(buf.data_ptr()), w * h * 2, n, interp, 1, 1, cs, lib.ptr(res), sp)
            torch.cuda.synchronize()
            assert rc == 0 and res.sum() == n, (rc, res)
        fill(); run()                                      # buffers, tables
        ts = []
        for r in range(reps):
            fill()
            t0 = time.perf_counter(); run(); ts.append(time.perf_counter() - t0)
        best, mean = min(ts), sum(ts) / len(ts)
        out[str(n)] = {"ms_per_batch": round(mean * 1e3, 2), "conversions_per_s": round(n / mean, 1), "best": round(n / best, 1)}
        sys.stderr.write(f"batch {n:3d}: {mean * 1e3:8.2f} ms per batch, {n / mean:7.1f} conversions/s (best {n / best:.1f})\n")
        del buf
        torch.cuda.empty_cache()
        if os.environ.get('DI_BENCH_TRIM', '1') == '1': L.mlvfs_amd_dualiso_trim()          # the next size starts from fresh work memory
finally:
    os.dup2(saved, 1)
print(json.dumps({"workload": f"{w}x{h} cr2hdr20 interp={interp} cs={cs} fullres alias-map, frames resident in HBM", "batch": out}))

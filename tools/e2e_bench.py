#!/usr/bin/env python3
"""PCIe-inclusive throughput of the hot path: 3584x1320 14-bit frames start and end in HOST memory.
  (a) mlvfs_amd_process_frames_host: pinned buffers, chunks of frames, H2D / kernels / D2H overlapped on three streams
  (b) the drop-in symbols exactly as MLVFS's process_frame calls them (one frame per call, synchronous, in place),
      from 1..T host threads (libfuse's worker pool)
This is never bench.py's `value` (that is the HBM-resident rate); DESIGN.md section 7 quotes these numbers."""
import ctypes as C, os, sys, time, threading
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mlvfs_amd import abi, lib, synth
from mlvfs_amd.stream import ClipStream

W, H = 3584, 1320
N = int(os.environ.get("E2E_FRAMES", "256"))
s = ClipStream(W, H)
L = s.L
base = s.synth_packed(8, seed=1)
s.analyse_first_frame(base, cs=5, bad_pix=1, stripes=True, rand_mode=1)
host_in = torch.empty((N, s.packed_stride), dtype=torch.uint8, pin_memory=True)
host_in.view(N // 8, 8, s.packed_stride)[:] = base.cpu()
host_out = torch.empty((N, s.out_stride), dtype=torch.uint8, pin_memory=True)
res = {}
for chunk in (2, 4, 8, 16):
    for cs, fp, st, name in ((5, True, True, "unpack+badpix+cs5x5+stripes"), (0, False, False, "unpack only")):
        s.process_host(host_in, host_out, cs=cs, fix_pixels=fp, stripes=st, chunk=chunk)
        t0 = time.perf_counter()
        for _ in range(3):
            s.process_host(host_in, host_out, cs=cs, fix_pixels=fp, stripes=st, chunk=chunk)
        dt = (time.perf_counter() - t0) / 3
        gbs = N * (s.packed_stride + s.out_stride) / dt / 1e9
        print(f"host pipeline  chunk={chunk:2d}  {name:30s} {N / dt:8.0f} fps  {N * W * H / dt / 1e6:9.0f} Mpix/s  {gbs:6.1f} GB/s over PCIe (both directions)", flush=True)
# pageable memory
pg_in, pg_out = host_in.clone(), torch.empty((N, s.out_stride), dtype=torch.uint8)
s.process_host(pg_in, pg_out, cs=5, fix_pixels=True, stripes=True, chunk=8)
t0 = time.perf_counter(); s.process_host(pg_in, pg_out, cs=5, fix_pixels=True, stripes=True, chunk=8); dt = time.perf_counter() - t0
print(f"host pipeline  chunk= 8  pageable buffers, cs5x5            {N / dt:8.0f} fps", flush=True)

# (b) drop-in symbols, one frame per call
frame = synth.normal_frame(W, H, seed=1, frame=0)
packed_np = np.concatenate([synth.pack14(frame).astype("<u2"), np.zeros(4, "<u2")])
def worker(nf, out_counts, idx):
    fh = abi.make_frame_headers(W, H, black=synth.BLACK, white=synth.WHITE)
    fh.file_hdr.fileGuid = 0x1234
    img = np.zeros(W * H, np.uint16)
    name = b"clip.MLV"
    for k in range(nf):
        L.dng_get_image_data(C.byref(fh), lib.ptr(packed_np), lib.ptr(img), 0, img.nbytes)
        L.fix_focus_pixels(C.byref(fh), lib.ptr(img), 0)
        L.fix_bad_pixels(C.byref(fh), lib.ptr(img), 0, 0)
        L.chroma_smooth(C.byref(fh), lib.ptr(img), 5)
        corr = L.stripes_get_correction(name)
        if not corr:
            corr = L.stripes_new_correction(name)
            L.stripes_compute_correction(C.byref(fh), corr, lib.ptr(img), 0, img.size)
        L.stripes_apply_correction(C.byref(fh), corr, lib.ptr(img), 0, img.size)
    out_counts[idx] = nf
devnull = os.open(os.devnull, os.O_WRONLY); saved = os.dup(1); os.dup2(devnull, 1)
try:
    worker(2, [0], 0)                                   # clip state (map, coefficients)
    for T in (1, 4, 16, 64):
        nf = max(4, 256 // T)
        counts = [0] * T
        th = [threading.Thread(target=worker, args=(nf, counts, i)) for i in range(T)]
        t0 = time.perf_counter()
        for t in th: t.start()
        for t in th: t.join()
        dt = time.perf_counter() - t0
        sys.stderr.write(f"drop-in symbols, {T:3d} host threads, process_frame order (5 calls per frame): {sum(counts) / dt:8.0f} fps\n")
finally:
    C.CDLL(None).fflush(None); os.dup2(saved, 1)

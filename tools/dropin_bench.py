#!/usr/bin/env python3
"""PCIe-inclusive rate of the five drop-in symbols called per frame in process_frame's order (main.c:942-997), 3584x1320, from
1 and N host threads (libfuse's worker pool), every thread on its own malloc'ed frame buffer.  Prints one JSON line.
MLVFS_AMD_RESIDENT=1 in the environment selects the mode in which a stage takes up the device copy the previous stage left
(DESIGN.md 7); DROPIN_BRACKET=1 puts mlvfs_amd_frame_begin / mlvfs_amd_frame_end where process_frame calls mlvfs_load_chunks /
mlvfs_close_chunks (what integration/mlvfs_amd_wrap.c does at link level for the C host, tools/dropin_bench_c.sh); the results
are checked against each other by the caller (bench.py) through the hash printed here.
usage: python tools/dropin_bench.py [threads] [frames_per_thread]"""
import ctypes as C, json, os, sys, threading, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mlvfs_amd import abi, lib, synth

W, H = 3584, 1320
T = int(sys.argv[1]) if len(sys.argv) > 1 else 16
NF = int(sys.argv[2]) if len(sys.argv) > 2 else 12
real_stdout = os.dup(1); os.dup2(2, 1)                 # the bad-pixel list goes to stdout like the reference's (cs.c:307-311)
L = lib.load()
packed_np = [np.concatenate([synth.pack14(synth.normal_frame(W, H, seed=1, frame=k)).astype("<u2"), np.zeros(4, "<u2")]) for k in range(2)]
name = b"dropin_bench.MLV"
last = {}


PINNED = os.environ.get("DROPIN_PINNED") == "1"     # frame buffers from mlvfs_amd_host_alloc (page-locked, pooled), one per frame
BRACKET = os.environ.get("DROPIN_BRACKET") == "1"   # the frame bracket around the stages (main.c:923 / 998 through the wrap shim)
REUSE = os.environ.get("DROPIN_REUSE") == "1"       # one pageable frame buffer per thread, reused (no fresh pages per frame)


def worker(nf, idx, counts, start, t_begin):
    fh = abi.make_frame_headers(W, H, black=synth.BLACK, white=synth.WHITE)
    fh.file_hdr.fileGuid = 0x1234
    keep = np.empty(W * H, np.uint16) if REUSE else None
    srcs = packed_np
    if PINNED:                                          # the packed frames where a reader would have put them: page-locked too
        L.mlvfs_amd_init(0)
        srcs, src_ptrs = [], []
        for a in packed_np:
            p = L.mlvfs_amd_host_alloc(a.nbytes)
            v = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint16)), shape=(a.size,))
            v[:] = a
            srcs.append(v); src_ptrs.append(p)
    for k in range(-2, nf):                             # two frames of warm-up per thread (its stream and buffers), then the clock
        if k == 0:
            if start.wait() == 0:
                t_begin[0] = time.perf_counter()
            start.wait()
        pin = None
        if PINNED:                                      # a buffer per frame from the library's pool, given back after the frame
            pin = L.mlvfs_amd_host_alloc(W * H * 2)
            img = np.ctypeslib.as_array(C.cast(pin, C.POINTER(C.c_uint16)), shape=(W * H,))
        else:
            img = keep if REUSE else np.empty(W * H, np.uint16)      # a fresh buffer per frame, like process_frame's malloc
        src = srcs[(idx + k) % 2]
        if BRACKET:
            L.mlvfs_amd_frame_begin()
        L.dng_get_image_data(C.byref(fh), lib.ptr(src), lib.ptr(img), 0, img.nbytes)
        L.fix_focus_pixels(C.byref(fh), lib.ptr(img), 0)
        L.fix_bad_pixels(C.byref(fh), lib.ptr(img), 0, 0)
        L.chroma_smooth(C.byref(fh), lib.ptr(img), 5)
        corr = L.stripes_get_correction(name)
        if not corr:
            corr = L.stripes_new_correction(name)
            L.stripes_compute_correction(C.byref(fh), corr, lib.ptr(img), 0, img.size)
        L.stripes_apply_correction(C.byref(fh), corr, lib.ptr(img), 0, img.size)     # sizes in pixels (main.c:996)
        if BRACKET:
            L.mlvfs_amd_frame_end()                     # the one download of the frame
        if (idx + k) % 2 == 1 and k >= nf - 2:
            last[idx] = img.copy() if (PINNED or REUSE) else img
        if pin:
            L.mlvfs_amd_host_free(pin)
    counts[idx] = nf
    if PINNED:
        for p in src_ptrs:
            L.mlvfs_amd_host_free(p)


C.CDLL(None).srand(1)
worker(1, 0, [0], threading.Barrier(1), [0.0])         # clip state (map, coefficients) from frame 0
res = {"resident": "bracket" if BRACKET else os.environ.get("MLVFS_AMD_RESIDENT", "0"), "frames_per_thread": NF, "pinned_frame_buffers": PINNED, "reused_frame_buffers": REUSE}
for t in (1, T):
    counts = [0] * t
    start, t_begin = threading.Barrier(t), [0.0]
    th = [threading.Thread(target=worker, args=(NF, i, counts, start, t_begin)) for i in range(t)]
    for x in th: x.start()
    for x in th: x.join()
    dt = time.perf_counter() - t_begin[0]
    res[f"fps_{t}_threads"] = round(sum(counts) / dt, 1)
ref = None
same = True
for v in last.values():
    if ref is None: ref = v
    same = same and np.array_equal(v, ref)
res["frame1_hash"] = synth.fnv1a(ref) if ref is not None else None
res["identical_between_threads"] = bool(same)
L.stripes_free_corrections()
C.CDLL(None).fflush(None)
os.dup2(real_stdout, 1)
print(json.dumps(res), flush=True)

"""tools/pnoise_bracket_bench.py -- process_frame's order with --fix-pattern-noise through the drop-in symbols inside a frame bracket
(what the wrap-linked MLVFS does): unpack, pattern noise or deflicker, bad pixels, cs5x5; 3584x1320, one thread, milliseconds per frame."""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mlvfs_amd import abi, lib, synth
gpu = lib.load(); gpu.mlvfs_amd_init(0)
w, h = 3584, 1320
f = synth.normal_frame(w, h, seed=1)
packed = np.ascontiguousarray(synth.pack_bits(f), np.uint16)
fh = abi.make_frame_headers(w, h, black=synth.BLACK, white=synth.WHITE)
img = np.empty((h, w), np.uint16)
devnull = os.open(os.devnull, os.O_WRONLY); saved = os.dup(1); os.dup2(devnull, 1)
def frame(pn):
    gpu.mlvfs_amd_frame_begin()
    gpu.dng_get_image_data(C.byref(fh), lib.ptr(packed), lib.ptr(img), 0, img.nbytes)
    if pn == 1: gpu.fix_pattern_noise(lib.ptr(img), w, h, synth.WHITE, 0)
    if pn == 2:                                                       # deflicker (main.c:895-906): histogram of every second pixel, its median
        hist = gpu.hist_create((1 << 14) + 1)
        gpu.hist_add(hist, C.c_void_p(img.ctypes.data + 2), (img.nbytes - 1) // 2, 1)
        gpu.hist_median(hist)
        gpu.hist_destroy(hist)
    gpu.fix_bad_pixels(C.byref(fh), lib.ptr(img), 0, 0)
    gpu.chroma_smooth(C.byref(fh), lib.ptr(img), 5)
    gpu.mlvfs_amd_frame_end()
res = {}
for pn in (0, 1, 2):
    for _ in range(3): frame(pn)
    t0 = time.perf_counter()
    for _ in range(20): frame(pn)
    res[pn] = (time.perf_counter() - t0) / 20 * 1e3
C.CDLL(None).fflush(None); os.dup2(saved, 1)
print(f"bracketed frame, one thread: {res[0]:.2f} ms; with fix_pattern_noise {res[1]:.2f} ms; with deflicker {res[2]:.2f} ms")

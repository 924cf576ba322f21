"""tools/lj92_dropin_bench.py -- lj92_open + lj92_decode + lj92_close (the reference decoder's own calls, exported by the library) on a
3584x1320 frame from the reference's encoder, host memory in and out, one thread: milliseconds per frame.  With oracle/_ref present
(not on the GPU box unless built) the reference decoder is timed beside it."""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mlvfs_amd import lib, synth
gpu = lib.load(); gpu.mlvfs_amd_init(0)
w, h = 3584, 1320
f = synth.normal_frame(w, h, seed=1)
q = np.ascontiguousarray(np.block([[f[0::2, 0::2], f[0::2, 1::2]], [f[1::2, 0::2], f[1::2, 1::2]]]))
try:
    from oracle.bindings import Reference
    ref = Reference()
    stream = ref.lj92_encode(q, 14)
except Exception as e:
    from oracle import lj92_testenc as enc
    ref = None
    stream = enc.encode(q, 6, 14)
buf = np.frombuffer(stream, np.uint8).copy()
out = np.zeros(w * h, np.uint16)
def once():
    hd = C.c_void_p(); a, b, c = C.c_int(), C.c_int(), C.c_int()
    assert gpu.lj92_open(C.byref(hd), C.c_void_p(buf.ctypes.data), buf.size, C.byref(a), C.byref(b), C.byref(c)) == 0
    assert gpu.lj92_decode(hd, C.c_void_p(out.ctypes.data), out.size, 0, None, 0) == 0
    gpu.lj92_close(hd)
for _ in range(3): once()
t0 = time.perf_counter()
for _ in range(20): once()
ms = (time.perf_counter() - t0) / 20 * 1e3
line = f"lj92_open + lj92_decode + lj92_close, {w}x{h}, {len(stream) * 8 / (w * h):.1f} bits/px: {ms:.2f} ms per frame"
if ref is not None:
    t0 = time.perf_counter(); ref.lj92_decode(stream); line += f"; the reference's decoder: {(time.perf_counter() - t0) * 1e3:.1f} ms"
print(line)

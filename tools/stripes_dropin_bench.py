"""stripes_compute_correction through the drop-in symbol (first frame of a clip in MLVFS's process_frame, main.c:976-984): a 3584x1320
frame in host memory, the dither drawn from the application's libc rand() stream.  ms per call, and the reference beside it when
oracle/_ref is there."""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mlvfs_amd import abi, lib, synth

L = lib.load(); assert L.mlvfs_amd_init(0) == 0
W, H = 3584, 1320
f = synth.normal_frame(W, H, seed=1)
fh = abi.make_frame_headers(W, H, bpp=14, black=synth.BLACK, white=synth.WHITE)
libc = C.CDLL(None)
t = []
for k in range(6):
    libc.srand(1)
    corr = L.stripes_new_correction(f"clip{k}.MLV".encode())
    t0 = time.perf_counter()
    L.stripes_compute_correction(C.byref(fh), corr, lib.ptr(f), 0, f.size)
    t.append(time.perf_counter() - t0)
    co = list(corr.contents.coeffficients)
print("stripes_compute_correction %dx%d: %s ms (first call first); coefficients %s" % (W, H, " ".join(f"{x * 1e3:.2f}" for x in t), co))
after = libc.rand()
try:
    from oracle import bindings
    if bindings.have_ref():
        R = bindings.Reference()
        libc.srand(1)
        t0 = time.perf_counter(); needed, coeffs = R.stripes_compute(f, synth.BLACK, synth.WHITE, reseed=False); t1 = time.perf_counter()
        print(f"reference: {(t1 - t0) * 1e3:.1f} ms, same coefficients: {list(coeffs) == co}, same rand() position: {libc.rand() == after}")
except Exception as e:  # noqa: BLE001
    print("reference not available:", e)

"""The animated GIF preview of a clip (gif.c:82-244 through main.c's create_preview: ten frames spread over the clip, deflickered,
downscaled): mlvfs_amd_mlv_gif_data against the reference's own code, a 3584x1320 clip of 24 frames in the page cache; ms per preview."""
import ctypes as C, os, sys, tempfile, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mlvfs_amd import lib, mlvfile, synth

W, H, N = 3584, 1320, 24
L = lib.load(); assert L.mlvfs_amd_init(0) == 0
frames = [synth.normal_frame(W, H, seed=6, frame=k) for k in range(4)]
tmp = tempfile.mkdtemp()
names = mlvfile.write_clip(os.path.join(tmp, "M07-0007.MLV"), [synth.pack_bits(frames[k % 4]).tobytes() for k in range(N)], W, H, black=synth.BLACK)


def gif_of(path):
    h = L.mlvfs_amd_mlv_open(path.encode(), 0)
    assert h
    from mlvfs_amd import abi
    fh = abi.FrameHeaders()
    assert L.mlvfs_amd_mlv_frame_headers(h, 0, C.byref(fh)) == 1
    size = L.mlvfs_amd_gif_size(C.byref(fh))
    buf = np.zeros(size, np.uint8)
    n = L.mlvfs_amd_mlv_gif_data(h, lib.ptr(buf), 0, size)
    L.mlvfs_amd_mlv_close(h)
    return buf[:n].tobytes()


t = []
for k in range(4):
    t0 = time.perf_counter(); g = gif_of(names[0]); t.append((time.perf_counter() - t0) * 1e3)
print(f"GIF preview of a {W}x{H} clip ({N} frames): " + " ".join(f"{x:.1f}" for x in t) + f" ms (first call first), {len(g)} bytes")
try:
    from oracle import bindings
    if bindings.have_ref():
        R = bindings.Reference()
        t0 = time.perf_counter(); want = R.gif(names[0]); t1 = time.perf_counter()
        print(f"reference: {(t1 - t0) * 1e3:.1f} ms, identical: {want == g}")
except Exception as e:  # noqa: BLE001
    print("reference not timed:", e)

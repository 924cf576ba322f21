"""Time the dual-ISO preview (hdr_convert_data drop-in, host buffers; and the device-resident form) on 3584x1320."""
import ctypes as C, sys, time, os
import numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from mlvfs_amd import abi, lib, synth
import torch
gpu = lib.load(); gpu.mlvfs_amd_init(0)
w, h = 3584, 1320
f = synth.dual_iso_frame(w, h)
ts = []
for _ in range(7):
    img = f.copy(); fh = abi.make_frame_headers(w, h, black=synth.BLACK, white=synth.WHITE)
    t0 = time.perf_counter(); r = gpu.hdr_convert_data(C.byref(fh), lib.ptr(img), 0, img.nbytes); ts.append(time.perf_counter() - t0)
    assert r == 1
ts = np.array(ts[2:]) * 1e3
print(f"hdr_convert_data {w}x{h} (host buffers): {ts.mean():.2f} ms/frame")
geom = lib.Geom(w, h, 14, synth.BLACK, synth.WHITE, 0, 0)
src = torch.from_numpy(f.view(np.int16)).cuda(); ts = []
for _ in range(7):
    t = src.clone(); torch.cuda.synchronize(); t0 = time.perf_counter()
    r = gpu.mlvfs_amd_hdr_preview_dev(C.byref(geom), C.c_void_p(t.data_ptr()), t.numel() * 2, None); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
ts = np.array(ts[2:]) * 1e3
print(f"mlvfs_amd_hdr_preview_dev (device-resident): {ts.mean():.2f} ms/frame")

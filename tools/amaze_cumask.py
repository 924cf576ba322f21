"""k_amaze on a stream restricted to a share of the CUs (hipExtStreamCreateWithCUMask): does a tile get faster when fewer tiles
are live, i.e. when the live tile planes (2 MB each) fit the memory-side cache?  Prints ms per launch plan and us per tile-CU."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mlvfs_amd import lib, synth
import torch
L = lib.load(); L.mlvfs_amd_init(0)
hip = C.CDLL("libamdhip64.so")
w, h = 4112, 2064                                          # 33 x 17 = 561 tiles
raw = torch.from_numpy(synth.amaze_plane(w, h, 1)).cuda()
out = [torch.empty((h, w), dtype=torch.float32, device="cuda") for _ in range(3)]
tiles = ((w + 16 + 127) // 128) * ((h + 16 + 127) // 128)
for share in (256, 192, 128, 96, 64, 32):
    words = (C.c_uint32 * 8)()
    # spread the enabled CUs over the XCDs: bit i of the mask is CU i in the runtime's numbering (XCD-interleaved)
    step = 256 / share
    on = {int(k * step) for k in range(share)}
    for i in on:
        words[i // 32] |= 1 << (i % 32)
    st = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(st), 8, words)
    assert rc == 0, rc
    def run():
        r = L.mlvfs_amd_amaze_demosaic_dev(C.c_void_p(raw.data_ptr()), w, h, *[C.c_void_p(o.data_ptr()) for o in out], st)
        assert r == 0
        hip.hipStreamSynchronize(st)
    run()
    import time
    ts = []
    for _ in range(4):
        t0 = time.perf_counter(); run(); ts.append(time.perf_counter() - t0)
    ms = min(ts) * 1e3
    print(f"{share:3d} CUs: {ms:7.3f} ms for {tiles} tiles -> {ms * 1e3 * share / tiles:7.1f} us of one CU per tile", flush=True)
    hip.hipStreamDestroy(st)

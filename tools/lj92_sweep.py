"""LJ92 decoder vs the oracle over random geometries, predictors, tables and batch compositions (debug aid).
usage: python tools/lj92_sweep.py [seed] [cases]"""
import sys
import numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from mlvfs_amd import lib, lj92, synth
from oracle import lj92_testenc as enc
from oracle.bindings import Oracle
o = Oracle()
lib.load().mlvfs_amd_init(0)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
bad = n = 0
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 60):
    xres = int(rng.integers(1, 200)) * 2
    yres = int(rng.integers(1, 120)) * 2
    batch = int(rng.integers(1, 5))
    streams, wants = [], []
    for b in range(batch):
        kind = str(rng.choice(["noise", "smooth", "flat", "ramp16"]))
        bits = 14
        if kind == "noise":
            img = rng.integers(0, 16384, (yres, xres)).astype(np.uint16)
        elif kind == "smooth":
            yy, xx = np.mgrid[0:yres, 0:xres]
            img = (2048 + 40 * xx + 25 * yy + rng.integers(-20, 20, (yres, xres))).clip(0, 16383).astype(np.uint16)
        elif kind == "flat":
            img = np.full((yres, xres), int(rng.integers(0, 16384)), np.uint16)
        else:
            bits = 16
            img = rng.integers(0, 65536, (yres, xres)).astype(np.uint16)
        pred = int(rng.choice([1, 0, 2, 3])) if bits == 16 else int(rng.integers(0, 8))
        # the JPEG's own shape only has to hold the same number of values (main.c:646-667)
        shapes = [(yres, xres)]
        if yres % 2 == 0: shapes.append((yres // 2, xres * 2))
        if xres % 2 == 0 and xres > 2: shapes.append((yres * 2, xres // 2))
        h, w = shapes[int(rng.integers(0, len(shapes)))]
        s = enc.encode(np.ascontiguousarray(img.reshape(h, w)), pred, bits, comment=b"sweep" if rng.random() < 0.3 else None, ramp=bool(rng.random() < 0.3))
        st, dec = o.lj92_decode(s)
        assert st == 0
        streams.append(s)
        wants.append(o.lj92_untile(dec, xres, yres))
    got = lj92.decode_frames(streams, xres, yres).cpu().numpy().view(np.uint16)
    for b in range(batch):
        n += 1
        if not np.array_equal(got[b], wants[b]):
            bad += 1
            print("MISMATCH", it, b, xres, yres, lj92.info(streams[b]), int((got[b] != wants[b]).sum()))
print(f"lj92 sweep: {n} streams, {bad} mismatches")

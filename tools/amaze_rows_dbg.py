"""debug: k_amaze_rows.hip (LDS row streaming) against k_amaze.hip plane by plane, and both against the oracle's three planes"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mlvfs_amd import lib, synth
from oracle.bindings import Oracle
import torch
gpu = lib.load(); gpu.mlvfs_amd_init(0)
o = Oracle()
T, TT, HALF = 160, 160 * 160, 160 * 80
TILE = 13 * TT + 13 * HALF
FULL = ["cfa", "green", "delsq", "dw0", "dw1", "vcd", "hcd", "vcdalt", "hcdalt", "cdsq", "dgv", "dgh", "hcd2"]
HALFP = ["hvwt", "dgrb0", "dgrb1", "delp", "delm", "rbint", "curv_h", "curv_v", "sqm", "sqp", "pmwt", "rbm", "rbp"]
ORDER = ["cfa", "dw1", "dw0", "hcdalt", "vcdalt", "hcd", "dgv", "dgh", "hcd2", "vcd", "cdsq", "delsq", "delp", "delm", "sqp", "sqm", "rbm", "rbp",
         "pmwt", "rbint", "hvwt", "curv_h", "curv_v", "green", "dgrb0", "dgrb1"]

def run(raw, mode):
    h, w = raw.shape
    d_raw = torch.from_numpy(raw).cuda()
    out = [torch.full((h, w), float("nan"), dtype=torch.float32, device="cuda") for _ in range(3)]
    tiles = ((w + 16 + 127) // 128) * ((h + 16 + 127) // 128)
    planes = torch.zeros(tiles * TILE, dtype=torch.float32, device="cuda")
    nfx, nfy = C.c_int(0), C.c_int(0)
    rc = gpu.mlvfs_amd_amaze_debug(C.c_void_p(d_raw.data_ptr()), w, h, *[C.c_void_p(t.data_ptr()) for t in out], mode,
                                   C.c_void_p(planes.data_ptr()), planes.numel(), C.byref(nfx), C.byref(nfy))
    assert rc == 0, lib.last_error()
    torch.cuda.synchronize()
    return [t.cpu().numpy() for t in out], planes.cpu().numpy(), nfx.value, nfy.value

def plane(block, name):
    if name in FULL:
        k = FULL.index(name); return block[k * TT:(k + 1) * TT].reshape(T, T)
    k = HALFP.index(name); return block[13 * TT + k * HALF:13 * TT + (k + 1) * HALF].reshape(T, 80)

def compare(w, h, seed, verbose=True):
    raw = synth.amaze_plane(w, h, seed)
    raw[::2, ::2] *= 1.0 + 0.5 * ((np.arange(w)[None, ::2] // 3) % 2)
    raw = raw.clip(0, 0xFFFFF).astype(np.float32)
    want = o.amaze_demosaic(raw)
    got0, pl0, _, _ = run(raw, 0)
    got1, pl1, nfx, nfy = run(raw, 1)
    bad0 = [int((g.view(np.uint32) != x.view(np.uint32)).sum()) for g, x in zip(got0, want)]
    bad1 = [int((g.view(np.uint32) != x.view(np.uint32)).sum()) for g, x in zip(got1, want)]
    tiles_x = (w + 16 + 127) // 128
    print(f"{w}x{h}: complete tiles {nfx}x{nfy}; k_amaze alone vs oracle {bad0}; with k_amaze_rows {bad1}", flush=True)
    if any(bad1) and verbose:
        k = int(np.argmax(bad1))
        ys, xs = np.nonzero(got1[k].view(np.uint32) != want[k].view(np.uint32))
        print("   plane", k, "rows", ys.min(), ys.max(), "cols", xs.min(), xs.max(), "tiles", sorted({((y + 16) // 128, (x + 16) // 128) for y, x in zip(ys[:2000], xs[:2000])})[:8])
        for ty in range(nfy):
            for tx in range(nfx):
                b0 = pl0[(ty * tiles_x + tx) * TILE:][:TILE]; b1 = pl1[(ty * nfx + tx) * TILE:][:TILE]
                rep = []
                for name in ORDER:
                    a, b = plane(b0, name), plane(b1, name)
                    if name == "green":      # the rows kernel keeps G at R/B sites only
                        yy, xx = np.mgrid[0:T, 0:T]; m = ((yy + xx) % 2 == 0) & (yy >= 8) & (yy < 152) & (xx >= 8) & (xx < 152)
                        d = (a.view(np.uint32) != b.view(np.uint32)) & m
                    else:
                        d = a.view(np.uint32) != b.view(np.uint32)
                    lo = 12 if name in FULL else 6
                    inner = d[12:148, lo:(148 if name in FULL else 74)]
                    if inner.any():
                        ys, xs = np.nonzero(d)
                        iy, ix = np.nonzero(inner)
                        rep.append(f"{name}: {int(inner.sum())} inner (all {int(d.sum())}) first inner at r{iy[0] + 12} c{ix[0] + lo}: legacy {a[iy[0] + 12, ix[0] + lo]:.6g} rows {b[iy[0] + 12, ix[0] + lo]:.6g}")
                if rep:
                    print(f"   tile ({ty},{tx}):"); [print("      " + r) for r in rep[:6]]
                    return False
    return not any(bad1)

if __name__ == "__main__":
    rng = np.random.default_rng(11)
    sizes = [(int(a), int(b)) for a, b in (s.split("x") for s in sys.argv[1:])] or (
        [(304, 304), (432, 304), (304, 432), (560, 432), (688, 560), (3584, 1320), (1920, 1080), (1920, 540), (3584, 660)] +
        [(int(rng.integers(70, 400)) * 4, int(rng.integers(280, 1300))) for _ in range(int(os.environ.get("NRANDOM", "30")))])
    ok = all([compare(w, h, w * 7 + h) for (w, h) in sizes])
    print("OK" if ok else "MISMATCH")

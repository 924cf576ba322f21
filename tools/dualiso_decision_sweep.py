"""The device-side decisions of the batched dual-ISO conversion against the checker over RANDOM material: ISO ratios, scene
brightness from deep shadows to mostly clipped, noise, black levels, which rows are the bright ones, GBRG starts, frames that are no
dual ISO at all, tiny and odd geometries -- mixed batches of one geometry, results and pixels frame by frame, the table caches
carried through the batch like the reference carries them through a clip (debug aid / parity sweep; tests hold a fixed subset).
usage: python tools/dualiso_decision_sweep.py [seed] [batches]"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mlvfs_amd import lib, synth
from oracle.bindings import Oracle
import torch

L = lib.load(); L.mlvfs_amd_init(0)
o = Oracle()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
NB = int(sys.argv[2]) if len(sys.argv) > 2 else 60


def random_frame(w, h, black):
    kind = rng.choice(["dual", "dual", "dual", "dual", "normal", "flat", "noise"])
    y, x = np.mgrid[0:h, 0:w]
    if kind == "normal":
        return synth.normal_frame(w, h, seed=int(rng.integers(1, 999)), black=black)
    if kind == "flat":
        return np.full((h, w), black + int(rng.integers(0, 3000)), np.uint16)
    if kind == "noise":
        return rng.integers(0, 16384, (h, w)).astype(np.uint16)
    ratio = int(rng.choice([2, 3, 4, 8, 16, 1]))
    scale = float(rng.choice([0.02, 0.1, 0.5, 1.0, 2.5, 8.0]))
    noise = int(rng.choice([0, 3, 16, 60]))
    phase = int(rng.integers(0, 4))                         # which rows are bright: (y + phase) % 4 >= 2
    fx, fy = rng.uniform(0.005, 0.05), rng.uniform(0.005, 0.05)
    scene = 20 + 1500 * scale * (0.5 + 0.5 * np.sin(fx * x) * np.cos(fy * y)) + 400 * scale * (((x // 48) + (y // 40)) & 1)
    cfa = (y & 1) * 2 + (x & 1)
    gain = np.choose(cfa, [0.6, 1.0, 1.0, 0.5])
    lin = scene * gain
    bright = ((y + phase) % 4) >= 2
    lin = np.where(bright, lin * ratio, lin)
    if noise:
        lin = lin + rng.integers(-noise, noise + 1, (h, w))
    clip_at = int(rng.choice([15200, 14000, 11000, 16383])) - black
    lin = np.minimum(lin, clip_at + rng.integers(0, 16, (h, w)))
    return np.clip(lin + black, 0, 16383).astype(np.uint16)


devnull = os.open(os.devnull, os.O_WRONLY); saved = os.dup(1); os.dup2(devnull, 1)
bad = n = conv = 0
try:
    for b in range(NB):
        interp = int(rng.choice([1, 1, 1, 0]))
        w = int(rng.integers(10, 120)) * 4 if interp == 0 else int(rng.integers(20, 240)) * 2
        h = int(rng.integers(40, 300))
        if interp == 0:
            w, h = max(w, 40), max(h, 40)
        black = int(rng.choice([2048, 2047, 1024, 512, 2049]))
        white = int(rng.choice([15000, 16383, 12000]))
        fr, am, cs = int(rng.integers(0, 2)), int(rng.integers(0, 2)), int(rng.choice([0, 0, 0, 2, 5]))
        nf = int(rng.integers(1, 6))
        frames = [random_frame(w, h, black) for _ in range(nf)]
        o.L.orc_dualiso_reset()
        want = [o.cr2hdr20(f, black, white, interp, fr, am, cs, reset=False) for f in frames]
        L.mlvfs_amd_dualiso_reset()
        t = torch.from_numpy(np.stack(frames).view(np.int16)).cuda()
        res = np.full(nf, -7, np.int32)
        geom = lib.Geom(w, h, 14, black, white, 0, 0)
        rc = L.mlvfs_amd_cr2hdr20_batch_dev(C.byref(geom), C.c_void_p(t.data_ptr()), w * h * 2, nf, interp, fr, am, cs, lib.ptr(res), None)
        torch.cuda.synchronize()
        got = t.cpu().numpy().view(np.uint16)
        for k in range(nf):
            n += 1
            r0, img, _ = want[k]
            if r0 == -1:
                ok = res[k] == 0
            else:
                ok = rc == 0 and res[k] == r0 and np.array_equal(got[k], img if r0 == 1 else frames[k])
            conv += int(r0 == 1)
            if not ok:
                bad += 1
                if os.environ.get("SWEEP_DEBUG") and bad <= 2:      # the checker's scalars next to the device's decisions (MLVFS_AMD_DI_DEBUG=1)
                    o.L.orc_dualiso_reset()
                    sc = o.cr2hdr20(frames[k], black, white, interp, fr, am, cs, want_scalars=True)[3]
                    sys.stderr.write(f"  checker scalars (rggb, is_bright, white, white_bright, a, b, corr_ev, white_darkened): {list(sc)}\n")
                    os.environ["MLVFS_AMD_DI_DEBUG"] = "1"
                    t1 = torch.from_numpy(frames[k].view(np.int16).copy()).cuda()
                    r1 = np.zeros(1, np.int32)
                    L.mlvfs_amd_dualiso_reset()
                    L.mlvfs_amd_cr2hdr20_batch_dev(C.byref(geom), C.c_void_p(t1.data_ptr()), w * h * 2, 1, interp, fr, am, cs, lib.ptr(r1), None)
                    torch.cuda.synchronize()
                    del os.environ["MLVFS_AMD_DI_DEBUG"]
                d = np.abs(got[k].astype(int) - (img if r0 == 1 else frames[k]).astype(int))
                sys.stderr.write(f"MISMATCH batch {b} frame {k}/{nf} w={w} h={h} interp={interp} black={black} white={white} fr={fr} am={am} cs={cs} "
                                 f"rc={rc} r={r0},{res[k]} ndiff={(d > 0).sum()} max={d.max()}\n")
finally:
    C.CDLL(None).fflush(None); os.dup2(saved, 1)
if bad and os.environ.get("SWEEP_STOP"):
    pass
sys.stderr.write(f"dual-ISO decision sweep: {n} frames in {NB} batches, {conv} converted by the checker, {bad} mismatches\n")
sys.exit(1 if bad else 0)

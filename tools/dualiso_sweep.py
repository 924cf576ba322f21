"""GPU vs oracle over random geometries / switches of the full dual-ISO conversion (debug aid)."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from mlvfs_amd import abi, lib, synth
from oracle.bindings import Oracle
gpu = lib.load(); gpu.mlvfs_amd_init(0)
o = Oracle()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
devnull = os.open(os.devnull, os.O_WRONLY); saved = os.dup(1); os.dup2(devnull, 1)
bad = 0; n = 0
try:
    for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 60):
        interp = int(rng.integers(0, 2))
        w = int(rng.integers(10, 160)) * 4 if interp == 0 else int(rng.integers(20, 320)) * 2
        h = int(rng.integers(40, 400))
        gbrg = int(rng.integers(0, 2))
        fr, am, cs = int(rng.integers(0, 2)), int(rng.integers(0, 2)), int(rng.choice([0, 0, 2, 3, 5]))
        f = synth.dual_iso_frame(w, h + 2, seed=int(rng.integers(1, 1000)))
        f = f[1:h + 1].copy() if gbrg else f[:h].copy()
        r0, want, lv0 = o.cr2hdr20(f, synth.BLACK, synth.WHITE, interp, fr, am, cs, reset=True)
        gpu.mlvfs_amd_dualiso_reset()
        fh = abi.make_frame_headers(w, h, black=synth.BLACK, white=synth.WHITE)
        got = f.copy()
        r1 = gpu.cr2hdr20_convert_data(C.byref(fh), lib.ptr(got), interp, fr, am, cs, 0)
        n += 1
        if r0 == -1:            # configuration the oracle refuses (AMaZE with w % 4): the library must refuse too
            ok = r1 == 0
        else:
            ok = r0 == r1 and np.array_equal(got, want)
        if not ok:
            bad += 1
            d = np.abs(got.astype(int) - want.astype(int))
            sys.stderr.write(f"MISMATCH w={w} h={h} interp={interp} gbrg={gbrg} fr={fr} am={am} cs={cs} r={r0},{r1} ndiff={(d>0).sum()} max={d.max()}\n")
finally:
    C.CDLL(None).fflush(None); os.dup2(saved, 1)
sys.stderr.write(f"dual-ISO sweep: {n} cases, {bad} mismatches\n")

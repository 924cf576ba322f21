"""tools/side_sweep.py [seed] [cases] -- fix_pattern_noise and hdr_convert_data (drop-in symbols) against the oracle over random
geometries and material: flat / noisy / edgy frames with clipped and black runs for the pattern-noise medians; dual-ISO frames with
columns of clipped bright and deep-shadow dark pixels (chains of rewritten rows) for the preview.  A debug aid beyond the test suite."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mlvfs_amd import abi, lib, synth
from oracle.bindings import Oracle
o = Oracle()
gpu = lib.load(); gpu.mlvfs_amd_init(0)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 60
BLACK, WHITE = synth.BLACK, synth.WHITE
devnull = os.open(os.devnull, os.O_WRONLY); saved = os.dup(1); os.dup2(devnull, 1)
bad_pn = bad_hp = n_hp = 0
try:
    for it in range(cases):
        # ---- pattern noise
        w, h = int(rng.integers(12, 700)) * 2, int(rng.integers(12, 400)) * 2
        kind = it % 4
        if kind == 0: f = synth.normal_frame(w, h, seed=int(rng.integers(1, 999)))
        elif kind == 1: f = synth.adversarial_frame(w, h, seed=int(rng.integers(1, 999)))
        elif kind == 2: f = (BLACK + rng.integers(0, 60, (h, w)) + 3000 * (rng.random((h, w)) < 0.02)).astype(np.uint16)       # flat with outliers: long runs
        else:
            base = np.cumsum(rng.integers(-300, 301, (h, w)), axis=1) % 12000 + BLACK                                           # edgy: short runs
            f = np.clip(base + 16000 * (rng.random((h, w)) < 0.01), 0, 16383).astype(np.uint16)
        want = o.fix_pattern_noise(f, WHITE)
        got = f.copy()
        gpu.fix_pattern_noise(lib.ptr(got), w, h, WHITE, 0)
        if not np.array_equal(got, want):
            bad_pn += 1
            sys.stderr.write(f"pattern noise MISMATCH case {it}: {w}x{h} kind {kind}, {(got != want).sum()} px\n")
        # ---- preview
        w, h = int(rng.integers(20, 500)) * 2, int(rng.integers(40, 400))
        f = synth.dual_iso_frame(w, h, seed=int(rng.integers(1, 999)))
        bright = (np.arange(h) % 4 >= 2)[:, None] & np.ones((1, w), bool)
        special = np.where(bright, 16383, BLACK - int(rng.integers(0, 80))).astype(np.uint16)
        m = np.zeros((h, w), bool)
        step = int(rng.integers(5, 17))
        m[:, 3:w:step] = rng.random((h, len(range(3, w, step)))) < rng.random()
        f = np.where(m, special, f).astype(np.uint16)
        ok, want, levels = o.hdr_preview(f, BLACK, WHITE)
        fh = abi.make_frame_headers(w, h, black=BLACK, white=WHITE)
        got = f.copy()
        r = gpu.hdr_convert_data(C.byref(fh), lib.ptr(got), 0, got.nbytes)
        n_hp += int(ok == 1)
        if r != ok or not np.array_equal(got, want if ok else f):
            bad_hp += 1
            sys.stderr.write(f"preview MISMATCH case {it}: {w}x{h} ret {r} vs {ok}, {(got != want).sum()} px\n")
finally:
    C.CDLL(None).fflush(None); os.dup2(saved, 1)
print(f"side sweep: {cases} cases each; pattern noise mismatches {bad_pn}; preview mismatches {bad_hp} ({n_hp} converted)")

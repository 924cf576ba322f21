"""Fused pipeline (k_frame + k_pixfix) vs oracle over random geometries / switches (debug aid)."""
import os, sys
import numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import torch
from mlvfs_amd import lib, synth
from mlvfs_amd.stream import ClipStream, to_numpy_u16
from oracle.bindings import Oracle
o = Oracle()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
devnull = os.open(os.devnull, os.O_WRONLY); saved = os.dup(1); os.dup2(devnull, 1)
bad = n = 0
try:
    for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 60):
        w = int(rng.integers(5, 300)) * 2 if rng.random() < 0.5 else int(rng.integers(1, 40)) * 16
        if rng.random() < 0.3: w = (w // 8) * 8 + 8
        h = int(rng.integers(8, 300))
        cs = int(rng.choice([0, 2, 3, 5, 5]))
        badpix = int(rng.integers(0, 3)); st = int(rng.integers(0, 2)) if w % 8 == 0 else 0
        kind = str(rng.choice(["normal", "adversarial", "colour_cast", "low_light"]))
        if rng.random() < 0.25:                                # now and then a frame of many tiles (the 5x5 kernel's per-tile state machines)
            w, h = int(rng.integers(30, 70)) * 16, int(rng.integers(200, 420))
        nf = int(rng.integers(2, 5))
        if kind in ("colour_cast", "low_light"):
            gen = getattr(synth, kind + "_frame")
            frames = [gen(w, h, seed=int(rng.integers(1, 999)) + k) for k in range(nf)]
        else:
            gen = getattr(synth, kind + "_frame")
            sd = int(rng.integers(1, 999))
            frames = [gen(w, h, seed=sd, frame=k) for k in range(nf)]
        s = ClipStream(w, h, 14, synth.BLACK, synth.WHITE, device=0)
        if rng.random() < 0.4: s.set_t16_layout(int(rng.integers(1, 3)))        # else decided from the first frame
        packed = s.upload_packed([synth.pack_bits(f) for f in frames])
        s.analyse_first_frame(packed, cs=cs, bad_pix=badpix, stripes=bool(st), rand_mode=1)
        got = to_numpy_u16(s.process(packed, cs=cs, fix_pixels=bool(badpix), stripes=bool(st)))
        pixels = o.detect_bad_pixels(frames[0], synth.BLACK, int(badpix == 2)) if badpix else None
        corr = None
        ok = True
        for k, f in enumerate(frames):
            img = o.apply_bad_pixels(f, synth.BLACK, pixels) if badpix else f
            if cs: img = o.chroma_smooth(img, synth.BLACK, cs)
            if st:
                if corr is None: corr = o.stripes_compute(img, synth.BLACK, synth.WHITE, frame_size=w * h * 14 // 8)
                img = o.stripes_apply(img, synth.BLACK, synth.WHITE, *corr)
            if not np.array_equal(got[k].reshape(h, w), img):
                ok = False
                g2 = got[k].reshape(h, w)
                ys, xs = np.nonzero(g2 != img)
                where = ", ".join(f"({x},{y}): {g2[y, x]} want {img[y, x]}" for y, x in list(zip(ys, xs))[:6])
                sys.stderr.write(f"MISMATCH w={w} h={h} cs={cs} badpix={badpix} st={st} kind={kind} frame={k} ndiff={len(ys)}  {where}\n")
        s.close()
        n += 1; bad += (not ok)
finally:
    os.dup2(saved, 1)
sys.stderr.write(f"frame sweep: {n} cases, {bad} mismatches\n")

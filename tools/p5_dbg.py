"""debug aid: k_frame_p5 (forced) against the oracle on one geometry.   usage: python tools/p5_dbg.py W H [kind] [badpix] [stripes]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MLVFS_AMD_KF_P5", "2")
from mlvfs_amd import synth
from mlvfs_amd.stream import ClipStream, to_numpy_u16
from oracle.bindings import Oracle
o = Oracle()
w, h = int(sys.argv[1]), int(sys.argv[2])
kind = sys.argv[3] if len(sys.argv) > 3 else "normal"
badpix = int(sys.argv[4]) if len(sys.argv) > 4 else 0
st = int(sys.argv[5]) if len(sys.argv) > 5 else 0
gen = getattr(synth, kind + "_frame")
frames = [gen(w, h, seed=5 + k) if kind in ("low_light", "colour_cast") else gen(w, h, seed=5, frame=k) for k in range(2)]
s = ClipStream(w, h, 14, synth.BLACK, synth.WHITE, device=0)
packed = s.upload_packed([synth.pack_bits(f) for f in frames])
s.analyse_first_frame(packed, cs=5, bad_pix=badpix, stripes=bool(st), rand_mode=1)
got = to_numpy_u16(s.process(packed, cs=5, fix_pixels=bool(badpix), stripes=bool(st)))
pixels = o.detect_bad_pixels(frames[0], synth.BLACK, int(badpix == 2)) if badpix else None
corr = None
for k, f in enumerate(frames):
    img = o.apply_bad_pixels(f, synth.BLACK, pixels) if badpix else f
    img = o.chroma_smooth(img, synth.BLACK, 5)
    if st:
        if corr is None: corr = o.stripes_compute(img, synth.BLACK, synth.WHITE, frame_size=w * h * 14 // 8)
        img = o.stripes_apply(img, synth.BLACK, synth.WHITE, *corr)
    g2 = got[k].reshape(h, w)
    ys, xs = np.nonzero(g2 != img)
    print(f"frame {k}: {len(ys)} px differ; rows {sorted(set(ys))[:12]} ... cols {sorted(set(xs))[:12]} ... max col {xs.max() if len(xs) else -1}")
    if len(ys):
        rows = np.bincount(ys, minlength=h); print("  per row (first 40):", rows[:40].tolist())
s.close()

"""Golden vectors for the LJ92 decoder, made by the REFERENCE's own lj92.c (oracle/_ref):
streams written by its encoder (predictor 6) and hand-made streams with predictors 0..7, each with the image the
reference's decoder returns.      python tests/golden/make_lj92_golden.py  ->  tests/golden/lj92_vectors.npz"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import lj92_testenc as enc  # noqa: E402
from oracle.bindings import Reference  # noqa: E402
from test_lj92 import images  # noqa: E402


def main():
    ref = Reference()
    out = {}
    for (w, h) in ((64, 48), (136, 72)):
        for name, img in images(w, h).items():
            key = f"{name}_{w}x{h}"
            streams = {"refenc": ref.lj92_encode(img, 14)}
            if (w, h) == (64, 48):
                for p in range(8):
                    streams[f"pred{p}"] = enc.encode(img, p, 14)
            for k, s in streams.items():
                st, dec = ref.lj92_decode(s)
                assert st == 0
                out[f"{key}_{k}_stream"] = np.frombuffer(s, np.uint8)
                out[f"{key}_{k}_image"] = dec
    np.savez_compressed(os.path.join(HERE, "lj92_vectors.npz"), **out)
    print("wrote", len(out) // 2, "vectors")


if __name__ == "__main__":
    main()

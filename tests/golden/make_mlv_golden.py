"""Golden vectors for the MLV index, made by the REFERENCE's own index.c (oracle/_ref):
for each synthetic clip of tests/test_mlv_reader.py::CLIPS the XREF block of get_new_index and the .IDX file that
get_index leaves beside the clip.      python tests/golden/make_mlv_golden.py  ->  tests/golden/mlv_index.npz"""
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
from mlvfs_amd import mlvfile  # noqa: E402
from oracle.bindings import Reference  # noqa: E402
from test_mlv_reader import CLIPS, H, W, payloads  # noqa: E402


def main():
    ref = Reference()
    out = {}
    for name in sorted(CLIPS):
        kw = dict(CLIPS[name])
        n = kw.pop("n")
        with tempfile.TemporaryDirectory() as d:
            names = mlvfile.write_clip(os.path.join(d, "M27-1337.MLV"), payloads(n), W, H, **kw)
            out[name + "_xref"] = np.frombuffer(ref.mlv_index(names[0]), np.uint8)
            assert ref.mlv_frame_count(names[0]) == n
            out[name + "_idx"] = np.frombuffer(open(names[0][:-3] + "IDX", "rb").read(), np.uint8)
    np.savez_compressed(os.path.join(HERE, "mlv_index.npz"), **out)
    print("wrote", sorted(out))


if __name__ == "__main__":
    main()

"""Golden vectors for the DNG header writer, made by the REFERENCE's own
dng_get_header_data (oracle/_ref/libmlvfs_ref.so, built from /root/reference by oracle/Makefile).

    python tests/golden/make_header_golden.py      ->  tests/golden/header_cases.npz

Per case k (mlvfs_amd.synth.header_case): the 592-byte frame_headers image passed in, fps_override, the
reel name, the bytes returned, the first 2048 bytes of the header (the rest is checked to be zero here) and
the frame_headers image afterwards (the reference rewrites the active area).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from mlvfs_amd import synth  # noqa: E402
from oracle.bindings import Reference  # noqa: E402

N, KEEP = 90, 2048


def main():
    ref = Reference()
    blob_in, blob_out, head, fps, base, nret = [], [], [], [], [], []
    for k in range(N):
        fh, f, b = synth.header_case(k)
        blob = np.frombuffer(bytes(fh), np.uint8)
        n, out, after = ref.header_data(blob, 0, 65536, f, b)
        assert not out[KEEP:].any()
        blob_in.append(blob); blob_out.append(after); head.append(out[:KEEP].copy()); fps.append(f); base.append(b); nret.append(n)
    np.savez_compressed(os.path.join(HERE, "header_cases.npz"), blob_in=np.array(blob_in), blob_out=np.array(blob_out),
                        head=np.array(head), fps=np.array(fps), base=np.array(base, dtype="S64"), nret=np.array(nret))
    print("wrote", N, "cases")


if __name__ == "__main__":
    main()

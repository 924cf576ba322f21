#!/usr/bin/env python3
"""Generate tests/golden/lzma_vectors.npz and gif_hashes.json FROM THE REFERENCE'S OWN CODE (oracle/_ref: LZMA/LzmaLib.c,
gif.c with main.c's get_image_data sliced in at build time).  Run in the build container only:

    python tests/golden/make_lzma_gif_golden.py

Committed: payloads made by the reference's encoder (also truncated / corrupted ones) with the return code of the reference's
decoder and the SHA-256 of what it produced; SHA-256 of the reference's GIF preview for clips the test regenerates from the
seeded generator.  No reference source text is stored."""
import hashlib
import json
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from mlvfs_amd import mlvfile, synth  # noqa: E402
from oracle.bindings import Reference  # noqa: E402
import test_lzma_gif as T  # noqa: E402

ref = Reference()
vec = {}
k = 0


def add(payload):
    global k
    rc, out = ref.lzma_uncompress(payload)
    vec[f"payload{k}"] = np.frombuffer(payload, np.uint8)
    vec[f"rc{k}"] = np.int32(rc)
    vec[f"sha{k}"] = np.str_(hashlib.sha256(out).hexdigest() if rc == 0 else "")
    k += 1


for kind in range(6):
    for props in ((5, 1 << 20, 3, 0, 2), (9, 1 << 22, 8, 4, 4), (3, 12288, 1, 1, 3)):
        data = T.sample_data(kind)
        if len(data) > 20000:
            data = data[:20000]                                     # keep the fixture small
        add(ref.lzma_payload(data, *props))
base = ref.lzma_payload(T.sample_data(5))
for cut in (1, 5, 6, 23, 400):
    if cut < len(base) - 14:
        add(base[:-cut])
rng = np.random.default_rng(11)
for _ in range(12):
    b = bytearray(base)
    b[int(rng.integers(4, len(b)))] ^= int(rng.integers(1, 256))
    add(bytes(b))
vec["count"] = np.int32(k)
np.savez_compressed(os.path.join(HERE, "lzma_vectors.npz"), **vec)

gif = {}
with tempfile.TemporaryDirectory() as d:
    for key, (w, h, n, black) in {"plain_256x136": (256, 136, 23, synth.BLACK), "odd_250x130": (250, 130, 7, synth.BLACK),
                                  "big_1920x1080": (1920, 1080, 11, synth.BLACK)}.items():
        frames = [synth.normal_frame(w, h, seed=6, frame=j, black=black) for j in range(n)]
        os.makedirs(os.path.join(d, key))
        names = mlvfile.write_clip(os.path.join(d, key, "M03-0003.MLV"), [synth.pack_bits(f).tobytes() for f in frames], w, h, black=black)
        gif[key] = hashlib.sha256(ref.gif(names[0])).hexdigest()
json.dump(gif, open(os.path.join(HERE, "gif_hashes.json"), "w"), indent=1)
print(k, "lzma vectors;", gif)

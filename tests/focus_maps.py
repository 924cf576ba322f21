"""The reference's focus-pixel maps as committed data (tests/golden/focus_maps.npz, made by tests/golden/make_focus_golden.py
from mlvfs/data/*.fpm): coordinates only, delta-coded."""
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def golden():
    return json.load(open(os.path.join(HERE, "golden", "focus_golden.json")))


def load(name: str) -> np.ndarray:
    """(n, 2) int32 sensor coordinates of map "<camera hex>_<raw w>x<raw h>", in the file's order."""
    g = golden()
    z = np.load(os.path.join(HERE, "golden", "focus_maps.npz"))
    return np.cumsum(z[g["alias"].get(name, name)].astype(np.int64), axis=0).astype(np.int32)


def write_fpm(directory, name: str) -> np.ndarray:
    """The map as the text file cs.c:369-385 reads ("%i %i" per entry) in `directory`; returns the coordinates."""
    xy = load(name)
    with open(os.path.join(str(directory), name + ".fpm"), "w") as f:
        f.write("".join(f"{x} \t {y}\n" for x, y in xy))
    return xy

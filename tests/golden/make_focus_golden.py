#!/usr/bin/env python3
"""tests/golden/focus_maps.npz + focus_golden.json FROM THE REFERENCE (run in the build container only).

The reference ships sixteen focus-pixel maps under mlvfs/data/ (four cameras x four raw geometries; cs.c:356-401 reads
"<cameraModel hex>_<raw width>x<raw height>.fpm" from the current directory).  This script reads them WHERE THEY LIE, runs
the reference's own fix_focus_pixels (oracle/_ref/libmlvfs_ref.so) on seeded frames -- the normal rule and the dual-ISO rule,
crop offsets 0 and non-zero (panPosX rounded up to 8, panPosY down to 2: cs.c:439-440) -- and commits DATA only:

    focus_maps.npz       the coordinate lists in file order, DELTA-coded (row 0 absolute, then differences to the previous entry:
                         the maps are regular grids, 607 572 entries take 10 KB that way), int16, identical lists stored
                         once, key "<camera>_<w>x<h>"; tests/focus_maps.py:load() undoes the coding and the aliasing
    focus_golden.json    per case: geometry, pan, rule, FNV-1a of the reference's output, number of pixels it changed

    python tests/golden/make_focus_golden.py
"""
import glob
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from mlvfs_amd import synth  # noqa: E402
from oracle.bindings import Reference  # noqa: E402

DATA = "/root/reference/mlvfs/data"
BLACK = synth.BLACK


def video_geometry(raw_w, raw_h):
    """A recording area inside the raw area the way crop-mode video has one: width a multiple of 16 (the fused kernel's fast
    path) or not, height even."""
    return (raw_w - 80) // 16 * 16, (raw_h - 30) // 2 * 2


def cases_for(camera, raw_w, raw_h):
    w, h = video_geometry(raw_w, raw_h)
    yield dict(camera=camera, raw_w=raw_w, raw_h=raw_h, w=w, h=h, pan=[0, 0], dual_iso=0, kind="normal")
    yield dict(camera=camera, raw_w=raw_w, raw_h=raw_h, w=w, h=h, pan=[35, 13], dual_iso=0, kind="normal")
    yield dict(camera=camera, raw_w=raw_w, raw_h=raw_h, w=w, h=h, pan=[35, 13], dual_iso=1, kind="dual_iso")
    yield dict(camera=camera, raw_w=raw_w, raw_h=raw_h, w=w - 6, h=h, pan=[8, 2], dual_iso=0, kind="normal")     # width not a multiple of 8


def frame_for(case):
    f = synth.dual_iso_frame if case["kind"] == "dual_iso" else synth.normal_frame
    return f(case["w"], case["h"], seed=31)


def main():
    ref = Reference()
    maps, alias, by_bytes = {}, {}, {}
    cases = []
    for path in sorted(glob.glob(os.path.join(DATA, "*.fpm"))):
        name = os.path.basename(path)[:-4]
        camera, geo = name.split("_")
        raw_w, raw_h = (int(v) for v in geo.split("x"))
        xy = np.loadtxt(path, dtype=np.int64).reshape(-1, 2)
        assert xy.min() >= 0 and xy.max() < 32768
        key = xy.tobytes()
        if key in by_bytes:
            alias[name] = by_bytes[key]
        else:
            by_bytes[key] = name
            maps[name] = np.concatenate([xy[:1], np.diff(xy, axis=0)]).astype(np.int16)
            assert np.array_equal(np.cumsum(maps[name].astype(np.int64), axis=0), xy)
        # every geometry of the densest camera, one geometry of each of the others
        if camera == "80000346" or geo == "1808x727":
            os.chdir(DATA)                                  # cs.c:369-370: relative to the current directory
            for case in cases_for(int(camera, 16), raw_w, raw_h):
                f = frame_for(case)
                out = ref.fix_focus_pixels(f, BLACK, case["dual_iso"], case["camera"], raw_w, raw_h, tuple(case["pan"]))
                case["map"] = name
                case["entries"] = int(len(xy))
                case["changed"] = int((out != f).sum())
                case["hash"] = synth.fnv1a(out)
                assert case["changed"] > 0
                cases.append(case)
    np.savez_compressed(os.path.join(HERE, "focus_maps.npz"), **maps)
    json.dump({"generator": "tests/golden/make_focus_golden.py", "source": "mlvfs/data/*.fpm of the reference (coordinates only)",
               "alias": alias, "cases": cases}, open(os.path.join(HERE, "focus_golden.json"), "w"), indent=1)
    print(len(maps), "distinct maps,", len(alias), "duplicates,", len(cases), "cases,",
          os.path.getsize(os.path.join(HERE, "focus_maps.npz")), "bytes")


if __name__ == "__main__":
    main()

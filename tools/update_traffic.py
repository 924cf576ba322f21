#!/usr/bin/env python3
"""tools/update_traffic.py PMC_SUMMARY FRAMES_PER_LAUNCH [SOURCE_NOTE] -- refresh profiles/traffic.json (what bench.py quotes as
roofline.traffic) from a tools/profile_round.sh summary: FETCH_SIZE x 2 (gfx950 correction, MI355X_MICROARCH.md) + WRITE_SIZE, both
in KiB per launch, and SQ_INSTS_VALU, for the headline kernel."""
import json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
path, frames = sys.argv[1], int(sys.argv[2])
# one pass of the hot path = k_frame_p (round 5: every tile whose packed medians are certain) + the list-mode k_frame behind it (the rest):
# the counters of both are summed, per launch of each (they are launched in pairs)
# (the first pass of a long launch is the streaming form k_frame_p5, of a short one k_frame_p: whichever the summary holds)
kernels = ["void mlv::k_frame_p5<false, 1>(mlv::FrameArgs, int, int, int, int, int)", "void mlv::k_frame_p<5, true, 1, false>(mlv::FrameArgs)",
           "void mlv::k_frame<5, true, 1, false>(mlv::FrameArgs)"]
vals, on = {}, False
per_kernel = {}
for ln in open(path):
    if not ln.startswith(" "):
        on = next((k for k in kernels if ln.strip().startswith(k[:40])), None)
        continue
    m = re.match(r"\s+(\S+)\s+n=\s*\d+\s+mean=\s*([0-9.]+)", ln)
    if on and m:
        vals[m.group(1)] = vals.get(m.group(1), 0.0) + float(m.group(2))
        per_kernel.setdefault(on, {})[m.group(1)] = float(m.group(2))
kernel = next(k for k in kernels if k in per_kernel)
tj_path = os.path.join(ROOT, "profiles", "traffic.json")
tj = json.load(open(tj_path))
by = (2 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024 / frames
tj["k_frame_bytes_per_frame"] = by
tj["ratio"] = by / tj["algorithmic_bytes_per_frame"]
tj["valu_insts_per_frame"] = int(vals["SQ_INSTS_VALU"] / frames)
tj["kernel"] = kernel
tj["kernels_summed"] = {k: {c: v for c, v in d.items() if c in ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU")} for k, d in per_kernel.items()}
rel = os.path.relpath(os.path.abspath(path), ROOT)
tj["source"] = f"{rel} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_INSTS_VALU, separate passes of `bench.py --no-cpu-baseline --no-extras`, " \
               f"{frames} frames per launch, tools/profile_round.sh" + (", " + sys.argv[3] if len(sys.argv) > 3 else "") + ")"
tj["valu_source"] = f"{rel} SQ_INSTS_VALU / {frames} frames (k_frame_p + list mode: 3238240, round 4: 3936284, round 3: 4293213, round 1: 4795296)"
json.dump(tj, open(tj_path, "w"), indent=1)
print(json.dumps({k: tj[k] for k in ("k_frame_bytes_per_frame", "ratio", "valu_insts_per_frame")}))

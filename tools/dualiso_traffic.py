#!/usr/bin/env python3
"""tools/dualiso_traffic.py PMC_GLOB KERNEL_STATS_CSV BENCH_LOG -- profiles/r04/dualiso_traffic.json: per kernel of a batch of 8 dual-ISO
conversions the HBM-side bytes (FETCH_SIZE doubled per the gfx950 correction + WRITE_SIZE, both reported in KiB), the vector
wave-instructions and the time; totals per frame; which kernel dominates.  The process runs three batches (one warm-up): sums are
divided by 8 frames x the number of batches (= launches of k_di_analyse)."""
import csv, glob, json, sys, collections, re
pmc = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.Counter()
seen_disp = set()
for path in glob.glob(sys.argv[1], recursive=True):
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        if "mlv::" not in k: continue
        k = re.sub(r"\(.*", "", k).replace("void ", "")
        pmc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (path, r.get("Dispatch_Id"), r["Counter_Name"])
        if r["Counter_Name"] == "SQ_INSTS_VALU": calls[k] += 1
stats = {}
for r in csv.DictReader(open(sys.argv[2])):
    k = r["Name"]
    if "mlv::" not in k: continue
    k = re.sub(r"\(.*", "", k).replace("void ", "")
    stats[k] = {"calls": int(r["Calls"]), "total_ns": float(r["TotalDurationNs"]), "avg_ns": float(r["AverageNs"])}
once = next((k for k in stats if "k_di_analyse" in k), None)           # launched once per batch (the frame index is in its grid)
nbatches = stats[once]["calls"] if once else 3
frames = 8 * nbatches
per_kernel = {}
tot_bytes = tot_valu = tot_ns = 0.0
for k, st in sorted(stats.items(), key=lambda kv: -kv[1]["total_ns"]):
    c = pmc.get(k, {})
    # counters were summed over the dispatches of the PMC passes, which launch the same kernels as the stats pass
    fetch_b = c.get("FETCH_SIZE", 0.0) * 1024 * 2          # gfx950: FETCH_SIZE tallies 128-byte requests at 64 bytes
    write_b = c.get("WRITE_SIZE", 0.0) * 1024
    valu = c.get("SQ_INSTS_VALU", 0.0)
    per_kernel[k] = {"launches_per_frame": round(st["calls"] / frames, 3), "us_per_frame": round(st["total_ns"] / frames / 1e3, 2),
                     "hbm_bytes_per_frame": int((fetch_b + write_b) / frames), "fetch_bytes_per_frame": int(fetch_b / frames),
                     "write_bytes_per_frame": int(write_b / frames), "valu_wave_insts_per_frame": int(valu / frames),
                     "achieved_GBps": round((fetch_b + write_b) / st["total_ns"], 1) if st["total_ns"] else None}
    tot_bytes += (fetch_b + write_b) / frames; tot_valu += valu / frames; tot_ns += st["total_ns"] / frames
dom = max(per_kernel.items(), key=lambda kv: kv[1]["us_per_frame"])
amaze_us = sum(v["us_per_frame"] for k, v in per_kernel.items() if "k_amaze" in k)
bench = {}
try:
    for ln in open(sys.argv[3]):
        if ln.startswith("{"): bench = json.loads(ln)["batch"]["8"]
except Exception: pass
out = {"workload": "batch of 8 conversions, 3584x1320 cr2hdr20 amaze-edge fullres alias-map, frames resident in HBM (tools/dualiso_batch_bench.py 8 2 under rocprofv3)",
       "frames_measured": frames,
       "traffic_bytes_per_frame": int(tot_bytes), "compulsory_bytes_per_frame": 17740800,
       "traffic_over_compulsory": round(tot_bytes / 17740800, 1),
       "valu_wave_insts_per_frame": int(tot_valu), "kernel_us_per_frame_sum": round(tot_ns / 1e3, 1),
       "dominant_kernel": dom[0], "dominant_kernel_us_per_frame": dom[1]["us_per_frame"],
       "amaze_share_of_kernel_sum": round(amaze_us / (tot_ns / 1e3), 3),
       "k_amaze_share_of_kernel_sum": round(sum(v["us_per_frame"] for k, v in per_kernel.items() if k.endswith("k_amaze")) / (tot_ns / 1e3), 3),
       "bench_under_profiler": bench, "per_kernel": per_kernel,
       "note": "FETCH_SIZE x 2 (gfx950 correction, MI355X_MICROARCH.md HBM section) + WRITE_SIZE, KiB -> bytes; kernels of two streams overlap, so the "
               "sum of kernel times exceeds the batch's wall time"}
print(json.dumps(out, indent=1))

#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: mean per dispatch, per kernel."""
import csv, sys, collections, glob
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sys.argv[1:]:
    for path in glob.glob(f, recursive=True):
        for r in csv.DictReader(open(path)):
            k = r["Kernel_Name"]
            if "mlv::" not in k: continue
            agg[k[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:28s} n={len(v):3d} mean={sum(v)/len(v):16.1f}")

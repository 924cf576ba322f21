#!/bin/bash
# tools/amaze_rows_stats.sh -- on the GPU box: rocprofv3 kernel statistics of the two AMaZE kernels in batches of 8 dual-ISO conversions
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/tmp_rows; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st -- python3 $R/tools/dualiso_batch_bench.py 8 3 > $O/stats.log 2>&1 || { tail -5 $O/stats.log; exit 1; }
find $O/st -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
rm -rf $O/st
python3 - <<PY
import csv
for r in csv.DictReader(open("$O/kernel_stats.csv")):
    if "amaze" in r["Name"] or "interp" in r["Name"]:
        print(f"{r['Name'][:34]:34s} calls {r['Calls']:>3s} avg {float(r['AverageNs'])/1e3:9.1f} us min {float(r['MinNs'])/1e3:9.1f} max {float(r['MaxNs'])/1e3:9.1f}  {r['Percentage']}%")
PY
grep batch $O/stats.log

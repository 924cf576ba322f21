#!/bin/bash
# tools/ab_di_stats.sh NAME... -- on the GPU box: per variant library build/ab/NAME.so the kernel statistics (rocprofv3 --kernel-trace --stats)
# of one batched dual-ISO bench run (tools/dualiso_batch_bench.py 8 3): average time per launch of every k_di_* / k_amaze* kernel
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/ab_di; mkdir -p $O
cp $R/mlvfs_amd/libmlvfs_amd.so $R/build/ab/_orig.so
cd /tmp && export TMPDIR=/tmp
for n in "$@"; do
  cp $R/build/ab/$n.so $R/mlvfs_amd/libmlvfs_amd.so
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st -- python3 $R/tools/dualiso_batch_bench.py 8 3 > $O/$n.log 2>&1 || echo "$n: the bench itself failed (experiments that change results do)"
  echo "== $n  $(grep -o '"8": {[^}]*}' $O/$n.log | head -1)"
  find $O/st -name "*kernel_stats.csv" -exec cp {} $O/$n.kernel_stats.csv \;
  rm -rf $O/st
  python3 - $O/$n.kernel_stats.csv <<'P'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Name"].split("(")[0].replace("void ", "").replace("mlv::", "")
    if k.startswith("k_"): print(f"   {k:28s} calls {int(r['Calls']):4d}  avg {float(r['AverageNs'])/1e3:9.1f} us  total {float(r['TotalDurationNs'])/1e6:8.2f} ms")
P
done
cp $R/build/ab/_orig.so $R/mlvfs_amd/libmlvfs_amd.so

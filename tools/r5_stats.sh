#!/bin/bash
# tools/r5_stats.sh KIND VARIANTS [P] -- on the GPU box: rocprofv3 kernel statistics of tools/kbench.py for one footage kind (normal | low_light | colour_cast)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
mkdir -p gpurun_out/r05
export TMPDIR=/tmp KB_KIND=$1 KB_ONLY=$2 MLVFS_AMD_KF_P=${3:-1} KB_ROUNDS=4 KB_FRAMES=${KB_FRAMES:-100}
rm -rf /tmp/st_$1
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st_$1 -- python3 $R/tools/kbench.py > /tmp/st_$1.log 2>&1)
f=$(find /tmp/st_$1 -name "*kernel_stats.csv" | head -1)
echo "== kind $1 variants $2 MLVFS_AMD_KF_P=$MLVFS_AMD_KF_P ($KB_FRAMES frames per launch)"
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "k_frame" in r["Name"]: print("%-60s calls %4s avg %10.1f us  total %8.2f ms" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY

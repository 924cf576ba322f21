#!/bin/bash
# tools/dualiso_traffic.sh [TAG] -- on the GPU box: what a batch of 8 dual-ISO conversions (BASELINE.json configs[3], 3584x1320,
# amaze-edge, full-res, alias map) really moves and executes, per kernel: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ counters in
# passes of their own (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do not fit one pass; no trace domain besides --kernel-trace),
# and a --kernel-trace --stats pass for the durations.  Output: gpurun_out/TAG/dualiso_traffic.json (+ the summaries it came from);
# copy to profiles/.  The script measures `tools/dualiso_batch_bench.py 8 2`: a warm-up batch and two measured ones per process.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; TAG=${1:-r04}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/dit_*
B="python3 $R/tools/dualiso_batch_bench.py 8 2"
p=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES"; do
  p=$((p + 1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/dit_pmc$p -- $B > /tmp/dit_pmc$p.log 2>&1 || { echo "pmc pass $p ($set) failed"; tail -3 /tmp/dit_pmc$p.log; }
done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/dit_stats -- $B > $O/dualiso_traffic_bench.log 2>&1
cd $R
python tools/pmc_summary.py "/tmp/dit_pmc*/**/*counter_collection.csv" > $O/dualiso_traffic_pmc_summary.txt
find /tmp/dit_stats -name "*kernel_stats.csv" -exec cp {} $O/dualiso_traffic_kernel_stats.csv \;
python tools/dualiso_traffic.py "/tmp/dit_pmc*/**/*counter_collection.csv" $O/dualiso_traffic_kernel_stats.csv $O/dualiso_traffic_bench.log > $O/dualiso_traffic.json
cat $O/dualiso_traffic.json | head -60

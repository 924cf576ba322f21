#!/bin/bash
# tools/kinds_pmc.sh -- on the GPU box: SQ counters of k_frame per footage kind (tools/kbench.py, 50 frames per launch), rocprofv3 --pmc
# in passes of their own.  Output: gpurun_out/kinds_pmc.log
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
mkdir -p gpurun_out
export TMPDIR=/tmp KB_ROUNDS=2
: > gpurun_out/kinds_pmc.log
for kind in normal colour_cast low_light; do
  export KB_KIND=$kind
  p=0
  for set in "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES"; do
    p=$((p + 1))
    rm -rf /tmp/kp_${kind}_$p
    (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/kp_${kind}_$p -- python3 $R/tools/kbench.py > /tmp/kp_${kind}_$p.log 2>&1) || { echo "pass $kind $p failed"; tail -5 /tmp/kp_${kind}_$p.log; exit 1; }
  done
  echo "== $kind" >> gpurun_out/kinds_pmc.log
  python tools/pmc_summary.py "/tmp/kp_${kind}_*/**/*counter_collection.csv" | grep -A13 "k_frame<[25]" >> gpurun_out/kinds_pmc.log
done
tail -120 gpurun_out/kinds_pmc.log

#!/bin/bash
# tools/amaze_rows_pmc.sh [TAG] -- on the GPU box: SQ / LDS / memory counters of k_amaze_rows in a batch of 8 (one rocprofv3 pass per set)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; TAG=${1:-r03}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/arp_*
p=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU" "FETCH_SIZE WRITE_SIZE" "SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_IFETCH SQ_ACTIVE_INST_ANY"; do
  p=$((p + 1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/arp_pmc$p -- python3 $R/tools/dualiso_batch_bench.py 8 1 > /tmp/arp_pmc$p.log 2>&1 || { echo "pmc pass $p ($set) failed"; tail -3 /tmp/arp_pmc$p.log; }
done
cd $R && python tools/pmc_summary.py "/tmp/arp_pmc*/**/*counter_collection.csv" > $O/amaze_rows_pmc_all.txt
grep -A22 "k_amaze_rows" $O/amaze_rows_pmc_all.txt | head -24 > $O/amaze_rows_pmc_summary.txt; cat $O/amaze_rows_pmc_summary.txt

#!/bin/bash
# tools/dualiso_batch_pmc.sh [TAG] -- on the GPU box: memory and SQ counters of the batched dual-ISO conversion (batch of 8), rocprofv3,
# one pass per counter set (no trace domains besides --kernel-trace)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; TAG=${1:-r03}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/dib_*
p=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_BUSY_CYCLES"; do
  p=$((p + 1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/dib_pmc$p -- python3 $R/tools/dualiso_batch_bench.py 8 1 > /tmp/dib_pmc$p.log 2>&1 || { echo "pmc pass $p ($set) failed"; tail -3 /tmp/dib_pmc$p.log; }
done
cd $R && python tools/pmc_summary.py "/tmp/dib_pmc*/**/*counter_collection.csv" > $O/dualiso_batch_pmc_summary.txt
grep -A14 "k_amaze" $O/dualiso_batch_pmc_summary.txt | head -18; grep -A14 "k_di_interp" $O/dualiso_batch_pmc_summary.txt | head -16

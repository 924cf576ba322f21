#!/bin/bash
# tools/dualiso_pmc.sh -- on the GPU box: kernel stats and SQ counters of the full dual-ISO conversion (tools/dualiso_bench.py), rocprofv3
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
mkdir -p gpurun_out/dualiso
export TMPDIR=/tmp
rm -rf /tmp/di_*
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/di_stats -- python3 $R/tools/dualiso_bench.py 0 6 > /tmp/di_stats.log 2>&1)
find /tmp/di_stats -name "*kernel_stats.csv" -exec cp {} gpurun_out/dualiso/kernel_stats.csv \;
p=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY" "SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU"; do
  p=$((p + 1))
  (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/di_pmc$p -- python3 $R/tools/dualiso_bench.py 0 3 > /tmp/di_pmc$p.log 2>&1) || { echo "pmc pass $p failed"; tail -3 /tmp/di_pmc$p.log; }
done
python tools/pmc_summary.py "/tmp/di_pmc*/**/*counter_collection.csv" > gpurun_out/dualiso/pmc_summary.txt
cut -d, -f1-4 gpurun_out/dualiso/kernel_stats.csv | cut -c1-110 | head -24
grep -A13 "k_amaze" gpurun_out/dualiso/pmc_summary.txt | head -16

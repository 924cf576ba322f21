#!/bin/bash
# tools/kf_pmc_split.sh TAG -- on the GPU box: where the wave cycles of k_frame go beyond vector issue (VERDICT r3 next #1b).
# Two rocprofv3 --pmc passes of the bench command (8 SQ slots each), summary of the k_frame rows into gpurun_out/TAG/
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-kf_split}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1"
rocprofv3 -L > $O/counters_available.txt 2>&1 || true
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $O/p1 -- $B > $O/p1.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/p2 -- $B > $O/p2.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_WAVES SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU --output-format csv -d $O/p3 -- $B > $O/p3.log 2>&1
cd $R
python tools/pmc_summary.py "$O/p1/**/*counter_collection.csv" "$O/p2/**/*counter_collection.csv" "$O/p3/**/*counter_collection.csv" > $O/pmc_all.txt 2>&1 || true
grep -A26 "k_frame" $O/pmc_all.txt > $O/pmc_k_frame.txt || true
rm -rf $O/p1 $O/p2 $O/p3
cat $O/pmc_k_frame.txt; tail -2 $O/p3.log

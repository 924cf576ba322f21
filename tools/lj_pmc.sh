#!/bin/bash
# tools/lj_pmc.sh TAG -- on the GPU box: wave-cycle split, instruction counts and HBM bytes of the LJ92 decode kernels
# (tools/lj92_bench.py, 32 frames per call), rocprofv3 --pmc passes -> gpurun_out/TAG/pmc_lj.txt, kernel statistics -> kernel_stats.csv
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-lj_pmc}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export LJ_BATCHES=32
B="python3 $R/tools/lj92_bench.py 32"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st -o lj -- $B > $O/st.log 2>&1 && cp $O/st/lj_kernel_stats.csv $O/kernel_stats.csv
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $O/p1 -- $B > $O/p1.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/p2 -- $B > $O/p2.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/p3 -- $B > $O/p3.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/p4 -- $B > $O/p4.log 2>&1
cd $R
python tools/pmc_summary.py "$O/p1/**/*counter_collection.csv" "$O/p2/**/*counter_collection.csv" "$O/p3/**/*counter_collection.csv" "$O/p4/**/*counter_collection.csv" > $O/pmc_lj.txt 2>&1 || true
rm -rf $O/p1 $O/p2 $O/p3 $O/p4 $O/st
grep -A20 "k_lj_decode\|k_lj_chunk_maps\|k_lj_rows" $O/pmc_lj.txt | head -80

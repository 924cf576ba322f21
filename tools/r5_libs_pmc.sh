#!/bin/bash
# tools/r5_libs_pmc.sh KIND VARIANT NAME... -- SQ counters of the k_frame kernels for each variant library build/ab/NAME.so on one box
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
mkdir -p gpurun_out/r05
KIND=$1; VAR=$2; shift; shift
LOG=gpurun_out/r05/${TAG:-libs}_pmc.log
: > $LOG
cp mlvfs_amd/libmlvfs_amd.so build/ab/_orig.so
export TMPDIR=/tmp KB_KIND=$KIND KB_ONLY=$VAR KB_ROUNDS=3 KB_FRAMES=${KB_FRAMES:-100}
for n in "$@"; do
  cp build/ab/$n.so mlvfs_amd/libmlvfs_amd.so
  rm -rf /tmp/pmc_l
  (cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT --output-format csv -d /tmp/pmc_l -- python3 $R/tools/kbench.py > /tmp/pmc_l.log 2>&1)
  echo "== $n $KIND $VAR" >> $LOG
  python tools/pmc_summary.py "/tmp/pmc_l/**/*counter_collection.csv" | grep -A8 "k_frame" >> $LOG
done
cp build/ab/_orig.so mlvfs_amd/libmlvfs_amd.so
cat $LOG

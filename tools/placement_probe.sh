#!/bin/bash
# tools/placement_probe.sh -- on the GPU box: the batch-of-8 dual-ISO conversion in a process that starts at 8 frames and in one that
# grows from 4 (8 % slower), with and without the split into parts, and the per-kernel times of the unsplit runs
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/placement; mkdir -p $O
cd /tmp && export TMPDIR=/tmp DI_BENCH_TRIM=0
for seq in 8 4,8; do
  echo "== sequence $seq, split"; timeout -k 10 200 python3 $R/tools/dualiso_batch_bench.py $seq 4 2>&1 >/dev/null | grep "batch   8"
  echo "== sequence $seq, unsplit"; MLVFS_AMD_DI_SPLIT=0 timeout -k 10 200 python3 $R/tools/dualiso_batch_bench.py $seq 4 2>&1 >/dev/null | grep "batch   8"
  MLVFS_AMD_DI_SPLIT=0 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/tr_$seq -- python3 $R/tools/dualiso_batch_bench.py $seq 4 > $O/tr_$seq.log 2>&1
  python3 $R/tools/trace_tail.py $(find $O/tr_$seq -name "*kernel_trace.csv") 3
  rm -rf $O/tr_$seq
done

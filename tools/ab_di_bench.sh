#!/bin/bash
# tools/ab_di_bench.sh NAME... -- on the GPU box: the batched dual-ISO bench (8 conversions per submission, 8 batches) per variant library
# build/ab/NAME.so, three rounds round-robin so that box drift shows; mean and best per run
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
cp mlvfs_amd/libmlvfs_amd.so build/ab/_orig.so
for rnd in 1 2 3; do
  for n in "$@"; do
    cp build/ab/$n.so mlvfs_amd/libmlvfs_amd.so
    echo "$n (round $rnd): $(timeout -k 10 200 python tools/dualiso_batch_bench.py ${AB_DI_SIZES:-8} 8 2>/dev/null | tail -1 | cut -c1-200)"
  done
done
cp build/ab/_orig.so mlvfs_amd/libmlvfs_amd.so

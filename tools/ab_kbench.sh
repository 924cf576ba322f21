#!/bin/bash
# tools/ab_kbench.sh NAME... -- on the GPU box: run tools/kbench.py once per variant library build/ab/NAME.so (fresh process each,
# the variant copied over the in-tree library), twice round-robin so that box drift shows
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
mkdir -p gpurun_out/ab
cp mlvfs_amd/libmlvfs_amd.so build/ab/_orig.so
for rnd in 1 2; do
  for n in "$@"; do
    cp build/ab/$n.so mlvfs_amd/libmlvfs_amd.so
    echo "== $n (round $rnd)" >> gpurun_out/ab/kbench.log
    KB_ROUNDS=${KB_ROUNDS:-6} timeout -k 10 200 python tools/kbench.py 2>gpurun_out/ab/err.$n.log | grep -E "^(m0|m2|m5) " >> gpurun_out/ab/kbench.log
    grep KF_TIMES gpurun_out/ab/err.$n.log | tail -3 >> gpurun_out/ab/kbench.log || true
  done
done
cp build/ab/_orig.so mlvfs_amd/libmlvfs_amd.so
cat gpurun_out/ab/kbench.log

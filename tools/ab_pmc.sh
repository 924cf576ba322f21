#!/bin/bash
# tools/ab_pmc.sh NAME... -- on the GPU box: SQ_INSTS_VALU / SQ_WAVE_CYCLES / SQ_WAIT_ANY per k_frame launch (100 frames) for each variant
# library build/ab/NAME.so (rocprofv3 --pmc in its own run, bench.py without extras)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
mkdir -p gpurun_out/ab
cp mlvfs_amd/libmlvfs_amd.so build/ab/_orig.so
export TMPDIR=/tmp
for n in "$@"; do
  cp build/ab/$n.so mlvfs_amd/libmlvfs_amd.so
  rm -rf /tmp/pmc_$n
  (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU --output-format csv -d /tmp/pmc_$n -- python3 $R/bench.py --steps 3 --warmup 1 --frames-per-step 100 --no-cpu-baseline --no-extras > /tmp/pmc_$n.log 2>&1)
  echo "== $n" >> gpurun_out/ab/pmc.log
  python tools/pmc_summary.py "/tmp/pmc_$n/**/*counter_collection.csv" | grep -A5 "k_frame<5" >> gpurun_out/ab/pmc.log
done
cp build/ab/_orig.so mlvfs_amd/libmlvfs_amd.so
cat gpurun_out/ab/pmc.log

#!/bin/bash
# tools/dropin_profile.sh MODE PINNED [threads] -- HIP API / kernel / copy statistics of the C drop-in bench (rocprofv3, no counters)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
MODE=${1:-2}; PIN=${2:-0}; T=${3:-16}
test -x build/dropin_bench_c || bash tools/dropin_bench_c.sh 1 1 > /dev/null
export TMPDIR=/tmp MLVFS_AMD_RESIDENT=$MODE
rm -rf /tmp/dp_prof
(cd /tmp && timeout -k 10 300 rocprofv3 --hip-trace --kernel-trace --memory-copy-trace --stats --output-format csv -d /tmp/dp_prof -- $R/build/dropin_bench_c $R/build/dropin_frame0.bin $R/build/dropin_frame1.bin $T 24 $PIN 2>&1 | grep '^{')
mkdir -p gpurun_out/dropin_prof_${MODE}_${PIN}
for f in $(find /tmp/dp_prof -name "*stats.csv"); do cp $f gpurun_out/dropin_prof_${MODE}_${PIN}/; done
for f in gpurun_out/dropin_prof_${MODE}_${PIN}/*hip_api_stats.csv gpurun_out/dropin_prof_${MODE}_${PIN}/*memory_copy_stats.csv; do echo "== $f"; head -12 $f | cut -c1-150; done

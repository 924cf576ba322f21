#!/bin/bash
# tools/dualiso_api_profile.sh -- HIP API statistics of the dual-ISO conversion with several frames in flight (tools/dualiso_mt_bench.py)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
export TMPDIR=/tmp
rm -rf /tmp/di_api
(cd /tmp && timeout -k 10 300 rocprofv3 --hip-trace --kernel-trace --memory-copy-trace --stats --output-format csv -d /tmp/di_api -- python3 $R/tools/dualiso_mt_bench.py 0 6 2>&1 | grep "dual-ISO")
mkdir -p gpurun_out/dualiso_api
for f in $(find /tmp/di_api -name "*stats.csv"); do cp $f gpurun_out/dualiso_api/; done
for f in gpurun_out/dualiso_api/*hip_api_stats.csv gpurun_out/dualiso_api/*domain_stats.csv; do echo "== $f"; head -14 $f | cut -c1-120; done

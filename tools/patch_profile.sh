#!/bin/bash
# tools/patch_profile.sh -- on the GPU box: rocprofv3 kernel statistics of tools/patch_sweep.py on the reference's densest focus-pixel map
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out/r03; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pp_stats
PS_SIZES=0 PS_ONLY=80000346_2592x1108 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp_stats -- python3 $R/tools/patch_sweep.py > $O/patch_profile.log 2>&1
find /tmp/pp_stats -name "*kernel_stats.csv" -exec cp {} $O/patch_kernel_stats.csv \;
cut -d, -f1-4 $O/patch_kernel_stats.csv | cut -c1-150 | head -12

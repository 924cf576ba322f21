#!/bin/bash
# tools/dualiso_batch_profile.sh [TAG] -- on the GPU box: the batched dual-ISO bench, then its rocprofv3 kernel statistics
R=$GRAFT_REPO_ROOT; TAG=${1:-r03}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R && timeout -k 10 600 python tools/dualiso_batch_bench.py 1,2,4,8,16 4 > $O/dualiso_batch_bench.json 2> $O/dualiso_batch_bench.log
cat $O/dualiso_batch_bench.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/distats -- python3 $R/tools/dualiso_batch_bench.py 8 3 > $O/dualiso_batch_stats.log 2>&1
find $O/distats -name "*kernel_stats.csv" -exec cp {} $O/dualiso_batch_kernel_stats.csv \;
rm -rf $O/distats
head -25 $O/dualiso_batch_kernel_stats.csv | cut -c1-160

#!/bin/bash
# tools/amaze_rows_ab.sh [TAG] -- on the GPU box: the batched dual-ISO bench with the complete AMaZE tiles through k_amaze.hip alone
# (MLVFS_AMD_AMAZE_ROWS=0) and through k_amaze_rows.hip (default), then the kernel statistics of a batch of 8 with the latter
R=$GRAFT_REPO_ROOT; TAG=${1:-r03}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
echo "--- k_amaze.hip alone" > $O/amaze_rows_ab.log
MLVFS_AMD_AMAZE_ROWS=0 timeout -k 10 300 python tools/dualiso_batch_bench.py 1,8,16 4 2>> $O/amaze_rows_ab.log > /dev/null || exit 1
echo "--- complete tiles through k_amaze_rows.hip" >> $O/amaze_rows_ab.log
timeout -k 10 300 python tools/dualiso_batch_bench.py 1,8,16 4 2>> $O/amaze_rows_ab.log > /dev/null || exit 1
cat $O/amaze_rows_ab.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/distats -- python3 $R/tools/dualiso_batch_bench.py 8 3 > $O/dualiso_batch_stats.log 2>&1 || exit 1
find $O/distats -name "*kernel_stats.csv" -exec cp {} $O/dualiso_batch_rows_kernel_stats.csv \;
rm -rf $O/distats
head -8 $O/dualiso_batch_rows_kernel_stats.csv | cut -c1-60,150-260

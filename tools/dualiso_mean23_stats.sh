#!/bin/bash
# rocprofv3 kernel statistics of batches of 8 dual-ISO conversions with the mean23 interpolator
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/tmp_m23; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st -- python3 $R/tools/dualiso_batch_bench.py 8 3 1 > $O/stats.log 2>&1 || { tail -5 $O/stats.log; exit 1; }
find $O/st -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
rm -rf $O/st
python3 $R/tools/print_stats.py $O/kernel_stats.csv > $O/st.txt; head -16 $O/st.txt; grep "batch " $O/stats.log | head -1

#!/bin/bash
# tools/pnoise_stats.sh -- on the GPU box: rocprofv3 kernel statistics of fix_pattern_noise through the drop-in symbol (tools/pnoise_bench.py)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pnoise; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
python3 $R/tools/pnoise_bench.py 2>&1 | tail -1
timeout -k 10 300 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $O/st -- python3 $R/tools/pnoise_bench.py > $O/st.log 2>&1
find $O/st -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
find $O/st -name "*memory_copy_stats.csv" -exec cp {} $O/memcpy_stats.csv \;
rm -rf $O/st
python3 $R/tools/print_stats.py $O/kernel_stats.csv; cat $O/memcpy_stats.csv | cut -c1-200

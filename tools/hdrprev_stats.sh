#!/bin/bash
# tools/hdrprev_stats.sh -- on the GPU box: rocprofv3 kernel statistics of the dual-ISO preview (tools/hdrprev_bench.py)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/hdrprev; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
python3 $R/tools/hdrprev_bench.py 2>&1 | tail -2
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st -- python3 $R/tools/hdrprev_bench.py > $O/st.log 2>&1
find $O/st -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
rm -rf $O/st
python3 $R/tools/print_stats.py $O/kernel_stats.csv

#!/bin/bash
# tools/secondary_stats.sh -- on the GPU box: tools/secondary_stats.py, then its rocprofv3 kernel statistics
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/secondary; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
python3 $R/tools/secondary_stats.py 2>&1 | grep "per call"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st -- python3 $R/tools/secondary_stats.py > $O/st.log 2>&1
find $O/st -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
rm -rf $O/st
python3 $R/tools/print_stats.py $O/kernel_stats.csv

R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/gantt; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for n in ${1:-8 16}; do
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/tr -- python3 $R/tools/dualiso_batch_bench.py $n 3 > $O/tr.log 2>&1
python3 $R/tools/trace_gantt.py $(find $O/tr -name "*kernel_trace.csv") > $O/gantt_$n.txt
rm -rf $O/tr
done

# tools/profile_round.sh TAG -- on the GPU box: GPU tests, default bench, rocprofv3 kernel stats and PMC passes of the bench command
set -e
R=$GRAFT_REPO_ROOT
TAG=${1:-r02}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1
python bench.py > $O/bench_default.log 2> $O/bench_default.err
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu-baseline --no-extras"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $B --steps 5 --warmup 1 > $O/stats.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- $B --steps 3 --warmup 1 > $O/pmc_fetch.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- $B --steps 3 --warmup 1 > $O/pmc_write.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $O/pmc_sq -- $B --steps 3 --warmup 1 > $O/pmc_sq.log 2>&1
cd $R
python tools/pmc_summary.py "$O/pmc_fetch/**/*counter_collection.csv" "$O/pmc_write/**/*counter_collection.csv" "$O/pmc_sq/**/*counter_collection.csv" > $O/pmc_summary.txt 2>&1 || true
find $O/stats -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
rm -rf $O/stats $O/pmc_fetch $O/pmc_write $O/pmc_sq
tail -3 $O/gpu_tests.log; tail -1 $O/bench_default.log | cut -c1-600; grep -A6 "k_frame" $O/pmc_summary.txt

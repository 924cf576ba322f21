set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/v18
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1
python bench.py > $O/bench_default.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 5 --warmup 1 --frames-per-step 100 --no-cpu-baseline > $O/stats.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --steps 3 --warmup 1 --frames-per-step 100 --no-cpu-baseline > $O/pmc_fetch.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/bench.py --steps 3 --warmup 1 --frames-per-step 100 --no-cpu-baseline > $O/pmc_write.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $O/pmc_sq -- python3 $R/bench.py --steps 3 --warmup 1 --frames-per-step 100 --no-cpu-baseline > $O/pmc_sq.log 2>&1
LJ_BATCHES=16 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/lj92 -- python3 $R/tools/lj92_bench.py 16 > $O/lj92.log 2>&1
cd $R
python tools/pmc_summary.py "$O/pmc_fetch/**/*counter_collection.csv" "$O/pmc_write/**/*counter_collection.csv" "$O/pmc_sq/**/*counter_collection.csv" > $O/pmc_summary.txt 2>&1 || true
tail -3 $O/gpu_tests.log; tail -1 $O/bench_default.log | cut -c1-400

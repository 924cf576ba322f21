cd $GRAFT_REPO_ROOT
for b in 16 32 64 128; do
  export MLVFS_AMD_ANALYSE_BAND=$b
  tools/amaze_rows_stats.sh > /dev/null 2>&1
  echo "band $b: $(python3 tools/print_stats.py gpurun_out/tmp_rows/kernel_stats.csv | grep 'analyse' | cut -c1-80) $(grep 'batch ' gpurun_out/tmp_rows/stats.log | head -1)"
done

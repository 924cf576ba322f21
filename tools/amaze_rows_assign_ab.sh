cd $GRAFT_REPO_ROOT
U=$(python3 -c "print(','.join(['1000']*18))")
N2="ffffffff06100c38,ffff06110d2a0718,ffffffff0d280309,ffffffffffff030c,ffffffff0932030b,ffffffff00000931,ffffffff11400d29,ffffffff030a0b21,ffffffff11420612,ffffffffffff0308,ffffffffffff0719,ffffffffffff071a,ffffffffffff1141,ffffffffffff0b20,ffffffffffff0930,ffffffffffff0b22,ffffffff0c680e82,ffffffff0a700352,ffffffff0a71014a,ffffffffffff0a72,ffffffff014b0e81,ffffffff0c790e80,ffffffff0c7a0351,ffffffff03500353,ffffffff01480354,ffffffff0c780149,ffffffff12880458,ffffffff12890459,ffffffff128a045a,ffffffff0760128b,ffffffff014c045c,ffffffffffff045b"
for v in uniform tuned1 tuned2; do
  export MLVFS_AMD_AMAZE_ROWS_COSTS= MLVFS_AMD_AMAZE_ROWS_ASSIGN=
  unset MLVFS_AMD_AMAZE_ROWS_COSTS MLVFS_AMD_AMAZE_ROWS_ASSIGN
  [ $v = uniform ] && export MLVFS_AMD_AMAZE_ROWS_COSTS=$U
  [ $v = tuned2 ] && export MLVFS_AMD_AMAZE_ROWS_ASSIGN=$N2
  tools/amaze_rows_stats.sh > /dev/null 2>&1
  echo "$v: $(python3 tools/print_stats.py gpurun_out/tmp_rows/kernel_stats.csv | grep 'amaze' | tr '\n' ' ') $(grep 'batch ' gpurun_out/tmp_rows/stats.log | head -1)"
done

#!/bin/bash
# timing experiment: the rows kernel alone with passes switched off (results wrong)
R=$GRAFT_REPO_ROOT; cd $R
for m in ${@:-0 3ffff 3fffe 2 4 8 10 20 40 80 100 200 400 800 1000 2000 4000 8000 10000 20000}; do
  MLVFS_AMD_AMAZE_ROWS_SKIP=$m timeout -k 10 120 python tools/amaze_rows_time.py 2>&1 | grep skip
done

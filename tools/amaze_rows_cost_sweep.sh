#!/bin/bash
cd $GRAFT_REPO_ROOT
base="2630 2060 2780 2620 1160 950 1160 2940 2550 890 2360 1590 1360 2970 3200 810 2610 2860"
for k in -650 0 600 1200 2000; do
  c=$(for v in $base; do echo -n "$((v + k)),"; done)
  echo "K=$k $(MLVFS_AMD_AMAZE_ROWS_COSTS=$c MLVFS_AMD_AMAZE_ROWS_SKIP=4 python tools/amaze_rows_time.py 2>&1 | grep skip)"
done
# uniform costs (pure item count balance)
c=$(for v in $base; do echo -n "1000,"; done)
echo "uniform $(MLVFS_AMD_AMAZE_ROWS_COSTS=$c MLVFS_AMD_AMAZE_ROWS_SKIP=4 python tools/amaze_rows_time.py 2>&1 | grep skip)"

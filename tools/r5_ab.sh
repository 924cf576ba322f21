#!/bin/bash
# tools/r5_ab.sh TAG [pmc] -- on the GPU box: tools/kbench.py (cs2x2 / cs3x3 / cs5x5 with pixel map and stripes) on the three footage kinds with
# the packed-once kernel off, always on and adaptive (MLVFS_AMD_KF_P=0 / 2 / 1, one library, one box), how many tiles it listed for k_frame, and -- with
# `pmc` -- SQ_INSTS_VALU / SQ_WAVE_CYCLES / SQ_WAIT_ANY / SQ_ACTIVE_INST_VALU of both kernels for cs5x5 and cs2x2
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
TAG=${1:-ab}
O=$R/gpurun_out/r05
mkdir -p $O
LOG=$O/${TAG}_kbench.log
: > $LOG
export KB_FRAMES=${KB_FRAMES:-100}
for kd in normal low_light colour_cast; do
  for p in 0 2 1; do
    echo "== kind $kd MLVFS_AMD_KF_P=$p" >> $LOG
    MLVFS_AMD_KF_P=$p MLVFS_AMD_KF_P_DEBUG=0 KB_KIND=$kd KB_ONLY=${KB_ONLY:-m2,m3,m5} timeout -k 10 200 python tools/kbench.py 2>/dev/null | grep "us/frame" >> $LOG
  done
  MLVFS_AMD_KF_P=2 MLVFS_AMD_KF_P_DEBUG=1 KB_ROUNDS=2 KB_KIND=$kd KB_ONLY=m2,m5 timeout -k 10 200 python tools/kbench.py 2>&1 | grep "listed" | sort | uniq -c >> $LOG
done
cat $LOG
if [ "$2" == "pmc" ]; then
  export TMPDIR=/tmp
  PL=$O/${TAG}_pmc.log
  : > $PL
  for p in 0 2; do
    for m in m5 m2; do
      export MLVFS_AMD_KF_P=$p KB_ONLY=$m KB_ROUNDS=3 KB_KIND=normal
      rm -rf /tmp/pmc_ab
      (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU --output-format csv -d /tmp/pmc_ab -- python3 $R/tools/kbench.py > /tmp/pmc_ab.log 2>&1)
      echo "== MLVFS_AMD_KF_P=$p $m ($KB_FRAMES frames per launch)" >> $PL
      python tools/pmc_summary.py "/tmp/pmc_ab/**/*counter_collection.csv" | grep -A5 "k_frame" >> $PL
    done
  done
  cat $PL
fi

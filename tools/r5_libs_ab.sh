#!/bin/bash
# tools/r5_libs_ab.sh "NAME[:ENV=V,...]"... -- on the GPU box: kbench rows (cs2x2 / cs3x3 / cs5x5, with pixel map and stripes) on the three footage
# kinds for each variant library build/ab/NAME.so, two rounds round-robin (the boxes of the pool differ by +-4 %: compare inside one run)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
mkdir -p gpurun_out/r05
LOG=gpurun_out/r05/${TAG:-libs}_ab.log
: > $LOG
cp mlvfs_amd/libmlvfs_amd.so build/ab/_orig.so
export KB_FRAMES=${KB_FRAMES:-100}
for rnd in 1 2; do
  for kd in ${KINDS:-normal low_light colour_cast}; do
    for spec in "$@"; do
      n=${spec%%:*}; envs=""
      if [[ "$spec" == *:* ]]; then envs=$(echo "${spec#*:}" | tr ',' ' '); fi
      cp build/ab/$n.so mlvfs_amd/libmlvfs_amd.so
      echo "== round $rnd kind $kd $spec" >> $LOG
      env $envs KB_KIND=$kd KB_ONLY=${KB_ONLY:-m2,m3,m5} timeout -k 10 200 python tools/kbench.py 2>/dev/null | grep "us/frame" >> $LOG
    done
  done
done
cp build/ab/_orig.so mlvfs_amd/libmlvfs_amd.so
cat $LOG

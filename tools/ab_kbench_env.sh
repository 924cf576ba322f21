#!/bin/bash
# tools/ab_kbench_env.sh "NAME[:ENV=V,ENV=V...]"... -- like ab_kbench.sh, each variant library build/ab/NAME.so with its own environment
# (MLVFS_AMD_KF_RUN / MLVFS_AMD_KF_SINGLES sweeps); two rounds, round-robin, so that box drift shows
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
mkdir -p gpurun_out/ab
LOG=gpurun_out/ab/kbench_env.log
cp mlvfs_amd/libmlvfs_amd.so build/ab/_orig.so
for rnd in 1 2; do
  for spec in "$@"; do
    n=${spec%%:*}; envs=""
    if [[ "$spec" == *:* ]]; then envs=$(echo "${spec#*:}" | tr ',' ' '); fi
    cp build/ab/$n.so mlvfs_amd/libmlvfs_amd.so
    echo "== $spec (round $rnd)" >> $LOG
    env $envs KB_ROUNDS=${KB_ROUNDS:-6} timeout -k 10 200 python tools/kbench.py 2>gpurun_out/ab/err.$n.log | grep -E "^(${KB_SHOW:-m0|m2|m3|m5}) " >> $LOG
    grep KF_TIMES gpurun_out/ab/err.$n.log | tail -3 >> $LOG || true
  done
done
cp build/ab/_orig.so mlvfs_amd/libmlvfs_amd.so
cat $LOG

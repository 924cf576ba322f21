#!/bin/bash
# tools/ab_bench.sh "NAME[:ENV=V,...]"... -- on the GPU box: the default bench.py headline (400 frames per step since round 4, preheated; `config.value_at_100_frames_per_step` beside it) once per variant
# library build/ab/NAME.so, two rounds round-robin; prints value / ms_per_step / roofline.frac per run
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
mkdir -p gpurun_out/ab
LOG=gpurun_out/ab/bench_ab.log
cp mlvfs_amd/libmlvfs_amd.so build/ab/_orig.so
for rnd in 1 2; do
  for spec in "$@"; do
    n=${spec%%:*}; envs=""
    if [[ "$spec" == *:* ]]; then envs=$(echo "${spec#*:}" | tr ',' ' '); fi
    cp build/ab/$n.so mlvfs_amd/libmlvfs_amd.so
    env $envs timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras --steps ${AB_STEPS:-40} --warmup 5 2>gpurun_out/ab/bench_err.$n.log | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        d = json.loads(ln); print('%-60s round $rnd: %.1f fps  %.4f ms/step  frac %.4f  us/frame(kernel) %.3f' % ('$spec', d.get('fps', 0), d['ms_per_step'], d['roofline']['frac'], 17740800 * 100 / d['roofline']['achieved'] / 1e9 * 1e6 / 100))
" >> $LOG
  done
done
cp build/ab/_orig.so mlvfs_amd/libmlvfs_amd.so
cat $LOG

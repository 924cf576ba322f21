#!/bin/bash
# tools/kfp_ablate.sh build | run -- where k_frame_p's vector instructions go, measured: variants of the library with one phase compiled out
# (-DKFP_EXP_NOLOAD / NOPREF / NOMED / NOOUT: results are wrong, counters and timing only), SQ_INSTS_VALU of each on the GPU box
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
V="full NOLOAD NOPREF NOMED NOOUT"
if [ "$1" == "build" ]; then
  mkdir -p build/ab
  cd mlvfs_amd/csrc
  BASE="-O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Wall -Wno-unused-function -I../../include -mllvm --amdgpu-sched-strategy=max-ilp -fno-slp-vectorize"
  OBJS=$(ls *.o | grep -v '^k_frame_p.o$')
  for v in $V; do
    D="-DKFP_HOIST_IL"; [ $v != full ] && D="-DKFP_HOIST_IL -DKFP_EXP_$v"
    /opt/rocm/bin/hipcc $BASE $D -c k_frame_p.hip -o ../../build/ab/kfp_$v.o &
  done
  wait
  for v in $V; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build/ab/kfp_$v.so ../../build/ab/kfp_$v.o $OBJS; rm ../../build/ab/kfp_$v.o; done
  ls -la ../../build/ab/kfp_*.so
  exit 0
fi
mkdir -p gpurun_out/r05
LOG=gpurun_out/r05/${2:-kfp}_ablate.log
: > $LOG
cp mlvfs_amd/libmlvfs_amd.so build/ab/_orig.so
export TMPDIR=/tmp MLVFS_AMD_KF_P=2 KB_ROUNDS=3 KB_KIND=normal KB_FRAMES=100
for m in ${KB_VARIANTS:-m5 m2}; do
  for v in $V; do
    cp build/ab/kfp_$v.so mlvfs_amd/libmlvfs_amd.so
    export KB_ONLY=$m
    rm -rf /tmp/pmc_abl
    (cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_SALU --output-format csv -d /tmp/pmc_abl -- python3 $R/tools/kbench.py > /tmp/pmc_abl.log 2>&1)
    echo "== $m $v: $(grep us/frame /tmp/pmc_abl.log | head -1)" >> $LOG
    python tools/pmc_summary.py "/tmp/pmc_abl/**/*counter_collection.csv" | grep -A4 "k_frame_p" >> $LOG
  done
done
cp build/ab/_orig.so mlvfs_amd/libmlvfs_amd.so
cat $LOG

#!/bin/bash
# tools/ab_build.sh NAME [extra hipcc flags for k_frame.hip] -- builds build/ab/NAME.so: the in-tree library with a variant k_frame.o
# (A/B runs on one GPU box: tools/ab_kbench.sh swaps the variants in one after the other)
set -e
cd "$(dirname "$0")/../mlvfs_amd/csrc"
NAME=$1; shift
OUT=../../build/ab; mkdir -p $OUT
BASE="-O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Wall -Wno-unused-function -I../../include"
KF=${KF_FLAGS--mllvm --amdgpu-sched-strategy=max-ilp -fno-slp-vectorize}
/opt/rocm/bin/hipcc $BASE $KF "$@" -c k_frame.hip -o $OUT/$NAME.k_frame.o
OBJS=$(ls *.o | grep -v '^k_frame.o$')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/$NAME.so $OUT/$NAME.k_frame.o $OBJS
rm $OUT/$NAME.k_frame.o
echo built $OUT/$NAME.so

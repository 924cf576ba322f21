#!/bin/bash
# tools/r5_build_variant.sh NAME [extra hipcc flags] -- build/ab/NAME.so: the in-tree library with k_frame.o, k_frame_p.o and k_frame_s.o compiled with the
# extra flags (-DKF_EXP_... switches of k_frame_dev.h); A/B on one box: tools/r5_libs_ab.sh
set -e
cd "$(dirname "$0")/../mlvfs_amd/csrc"
NAME=$1; shift
OUT=../../build/ab; mkdir -p $OUT
BASE="-O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Wall -Wno-unused-function -I../../include -mllvm --amdgpu-sched-strategy=max-ilp -fno-slp-vectorize"
/opt/rocm/bin/hipcc $BASE "$@" -c k_frame.hip -o $OUT/$NAME.k_frame.o &
/opt/rocm/bin/hipcc $BASE "$@" -c k_frame_p.hip -o $OUT/$NAME.k_frame_p.o &
/opt/rocm/bin/hipcc $BASE "$@" -c k_frame_s.hip -o $OUT/$NAME.k_frame_s.o &
wait
OBJS=$(ls *.o | grep -v '^k_frame.o$' | grep -v '^k_frame_p.o$' | grep -v '^k_frame_s.o$')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/$NAME.so $OUT/$NAME.k_frame.o $OUT/$NAME.k_frame_p.o $OUT/$NAME.k_frame_s.o $OBJS
rm $OUT/$NAME.k_frame.o $OUT/$NAME.k_frame_p.o $OUT/$NAME.k_frame_s.o
echo built $OUT/$NAME.so

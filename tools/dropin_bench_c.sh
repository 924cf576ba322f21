#!/bin/bash
# tools/dropin_bench_c.sh [threads] [frames per thread] -- builds tools/dropin_bench_c.c against the in-tree library twice (as it
# is; and with integration/mlvfs_amd_wrap.c + --wrap of the chunk calls = the frame bracket) and runs MLVFS_AMD_RESIDENT=0, =1 and
# the wrapped build, with malloc'ed and with pooled page-locked buffers.  Two JSON lines per run (stderr of the program): the rates, and -- for the
# first run on a node with several GPUs -- which device every worker was bound to with the frames per second of each device.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
T=${1:-16}; N=${2:-16}
mkdir -p build
LINK="-L mlvfs_amd -lmlvfs_amd -Wl,-rpath,$R/mlvfs_amd -L/opt/rocm/lib -Wl,-rpath,/opt/rocm/lib -lamdhip64 -lm"
gcc -std=gnu99 -O2 -pthread -I include tools/dropin_bench_c.c tests/c_host_chunks.c -o build/dropin_bench_c $LINK || exit 1
gcc -std=gnu99 -O2 -pthread -I include tools/dropin_bench_c.c tests/c_host_chunks.c integration/mlvfs_amd_wrap.c \
    -Wl,--wrap=mlvfs_load_chunks -Wl,--wrap=mlvfs_close_chunks -o build/dropin_bench_c_wrap $LINK || exit 1
gcc -std=gnu99 -O2 -pthread -I include tools/dropin_bench_c.c tests/c_host_chunks.c integration/mlvfs_amd_wrap.c integration/mlvfs_amd_wrap_alloc.c \
    -Wl,--wrap=mlvfs_load_chunks -Wl,--wrap=mlvfs_close_chunks -Wl,--wrap=malloc -Wl,--wrap=calloc -Wl,--wrap=realloc -Wl,--wrap=free \
    -o build/dropin_bench_c_wrap_alloc $LINK || exit 1
python - <<'PY'
import sys, numpy as np
sys.path.insert(0, ".")
from mlvfs_amd import synth
for k in range(2):
    synth.pack14(synth.normal_frame(3584, 1320, seed=1, frame=k)).astype("<u2").tofile(f"build/dropin_frame{k}.bin")
PY
for pinned in 0 1; do
  for mode in 0 1; do
    MLVFS_AMD_RESIDENT=$mode timeout -k 10 300 build/dropin_bench_c build/dropin_frame0.bin build/dropin_frame1.bin $T $N $pinned 2>&1 | grep '^{'
  done
  timeout -k 10 300 build/dropin_bench_c_wrap build/dropin_frame0.bin build/dropin_frame1.bin $T $N $pinned 2>&1 | grep '^{' | sed 's/"resident": "0"/"resident": "wrap"/'

done
# the source unchanged (malloc / free), both shims in the link: process_frame's buffers come from the library's page-locked pool
timeout -k 10 300 build/dropin_bench_c_wrap_alloc build/dropin_frame0.bin build/dropin_frame1.bin $T $N 0 2>&1 | grep '^{' | sed 's/"resident": "0"/"resident": "wrap+alloc shim"/'

#!/bin/bash
# tools/ab_kinds.sh NAME... -- on the GPU box: tools/kbench.py on the benchmark's frames and on the two other footage kinds, per variant library
# (AB_KINDS="normal colour_cast" picks the kinds, AB_ROWS="m2|m5" the kbench rows)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
cp mlvfs_amd/libmlvfs_amd.so build/ab/_orig.so
for lib in "$@"; do
  cp build/ab/$lib.so mlvfs_amd/libmlvfs_amd.so
  for kind in ${AB_KINDS:-normal low_light colour_cast}; do
    echo "== $lib $kind"
    KB_KIND=$kind KB_ROUNDS=${KB_ROUNDS:-5} timeout -k 10 200 python tools/kbench.py 2>/dev/null | grep -E "^(${AB_ROWS:-m2|m3|m5}) "
  done
done
cp build/ab/_orig.so mlvfs_amd/libmlvfs_amd.so

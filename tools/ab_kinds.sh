for lib in cur chain4; do
  cp build/ab/$lib.so mlvfs_amd/libmlvfs_amd.so
  for kind in low_light colour_cast; do
    echo "== $lib $kind"
    KB_KIND=$kind KB_ROUNDS=5 timeout -k 10 200 python tools/kbench.py 2>/dev/null | grep -E "^(m2|m3|m5) "
  done
done

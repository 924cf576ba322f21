#!/bin/bash
# tools/hwq_probe.sh -- on the GPU box: batches of 8 / 16 / 32 dual-ISO frames by the size of the parts a batch is launched in, and by the
# order in which the process met the batch sizes (which decides the streams' creation order)
R=$GRAFT_REPO_ROOT; cd /tmp; export DI_BENCH_TRIM=0
for part in 4 2 3 5 8; do for seq in 8,16,32 4,8; do
  echo "parts of $part, sizes $seq:"; MLVFS_AMD_DI_PART=$part timeout -k 10 200 python3 $R/tools/dualiso_batch_bench.py $seq 4 2>&1 >/dev/null | grep -E "batch +(8|16|32)"
done; done

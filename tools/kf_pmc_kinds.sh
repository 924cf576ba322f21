#!/bin/bash
# tools/kf_pmc_kinds.sh TAG [KIND...] -- on the GPU box: what the footage kinds (tools/kbench.py KB_KIND) change in k_frame's counters.
# Per kind three rocprofv3 --pmc passes of `kbench.py` restricted to the cs2x2 and cs5x5 rows; the k_frame rows go to gpurun_out/TAG/KIND.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-kf_kinds}; shift
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export KB_ONLY=${KB_ONLY:-m2,m5} KB_ROUNDS=3
for kind in ${@:-normal low_light colour_cast}; do
  export KB_KIND=$kind
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM --output-format csv -d $O/p1 -- python3 $R/tools/kbench.py > $O/$kind.p1.log 2>&1 &&
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_BRANCH --output-format csv -d $O/p2 -- python3 $R/tools/kbench.py > $O/$kind.p2.log 2>&1 &&
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr TCP_TA_DATA_STALL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum --output-format csv -d $O/p3 -- python3 $R/tools/kbench.py > $O/$kind.p3.log 2>&1 || { echo "profiler pass failed for $kind"; tail -5 $O/$kind.p*.log; exit 1; }
  (cd $R && python tools/pmc_summary.py "$O/p1/**/*counter_collection.csv" "$O/p2/**/*counter_collection.csv" "$O/p3/**/*counter_collection.csv" | grep -A30 "k_frame<" > $O/$kind.txt)
  rm -rf $O/p1 $O/p2 $O/p3
  echo "== $kind"; cat $O/$kind.txt
done

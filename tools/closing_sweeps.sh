#!/bin/bash
# tools/closing_sweeps.sh -- on the GPU box: the random parity sweeps beyond the test suite, one after the other (progress lines to
# gpurun_out/closing_sweeps.log): side paths, dual-ISO conversions, batched decisions, the fused pipeline, AMaZE geometries
R=$GRAFT_REPO_ROOT; cd $R; L=$R/gpurun_out/closing_sweeps.log; : > $L
for s in 11 12 13; do timeout -k 10 400 python tools/side_sweep.py $s 60 2>&1 | tail -1 >> $L; done
for s in 21 22; do timeout -k 10 600 python tools/dualiso_sweep.py $s 60 2>&1 | tail -1 >> $L; done
for s in 31 32; do timeout -k 10 600 python tools/dualiso_decision_sweep.py $s 60 2>&1 | tail -1 >> $L; done
for s in 41 42; do timeout -k 10 400 python tools/frame_sweep.py $s 80 2>&1 | tail -1 >> $L; done
NRANDOM=60 timeout -k 10 600 python tools/amaze_rows_dbg.py 2>&1 | tail -2 >> $L
cat $L

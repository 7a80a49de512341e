#!/bin/bash
# round 5: wall ms/frame in the latency-bound modes (one rank of eight with batches of 8, frame by frame on one GPU) per environment setting
TAG=$1; shift; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
for V in "$@"; do
  [ "$V" == "-" ] && V="A=0"
  echo "=== [$V]" | tee -a $OUT/modes.log
  for i in 1 2; do env $V timeout -k 10 120 python3 tools/wall_batch.py 8 8 2>&1 | tail -1 | tee -a $OUT/modes.log; done
  for i in 1 2; do env $V timeout -k 10 120 python3 tools/wall_batch.py 1 1 2>&1 | tail -1 | tee -a $OUT/modes.log; done
  env $V timeout -k 10 120 python3 tools/wall_batch.py 2 8 2>&1 | tail -1 | tee -a $OUT/modes.log
done

#!/bin/bash
# round 5: where a wave's cycles go in the PRODUCTION closest-hit kernels (RT_TRACE_TIMING=1: scalar s_memtime stamps, same occupancy), per environment setting
TAG=$1; shift; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
for V in "$@"; do
  [ "$V" == "-" ] && V="A=0"
  echo "=== [$V]" | tee -a $OUT/timing.log
  env $V RT_TRACE_TIMING=1 timeout -k 10 300 python3 tools/prof_frames.py --frames 16 --batch 8 2>&1 | grep "trace timing\|^{" | cut -c1-420 | tee -a $OUT/timing.log
done

#!/bin/bash
# round 5: traversal statistics (instrumented kernels, RT_TRACE_STATS=1) per environment setting:  tools/r05_stats.sh <tag> "VAR=val" ...
TAG=$1; shift; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
for V in "$@"; do
  [ "$V" == "-" ] && V="A=0"
  echo "=== [$V]" | tee -a $OUT/stats.log
  env $V RT_LANES=1 RT_TRACE_STATS=1 timeout -k 10 300 python3 tools/prof_frames.py --frames 16 --batch 8 2>&1 | grep "trace stats" | grep -v bounce_shadow | cut -c1-700 | tee -a $OUT/stats.log
done

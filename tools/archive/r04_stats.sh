#!/bin/bash
# round 4: traversal statistics of the any-hit launch WITHOUT the AO rays (RT_PACKET_AO=1 takes them out) and of everything (default)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-r04st}; mkdir -p $OUT; cd $R
for V in "RT_PACKET_AO=1" "A=0"; do
  echo "=== [$V]" | tee -a $OUT/stats.log
  env $V RT_LANES=1 RT_TRACE_STATS=1 timeout -k 10 300 python3 tools/prof_frames.py --frames 16 --batch 8 2>&1 | grep "trace stats" | grep -v bounce_shadow | cut -c1-700 | tee -a $OUT/stats.log
done

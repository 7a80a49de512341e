#!/bin/bash
TAG=${1:-r03q}; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
timeout -k 10 400 env RT_QUAD_REFILL=1 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q 2>&1 | tail -1
run() { echo -n "[$*] " | tee -a $OUT/q.log; for i in 1 2 3; do env "$@" timeout -k 10 120 python3 tools/wall_batch.py 1 8 2>&1 | tail -1 | awk '{printf "%s ", $5}' | tee -a $OUT/q.log; done; echo | tee -a $OUT/q.log; }
run A=0
run RT_QUAD_REFILL=1
run RT_QUAD_REFILL=1 RT_REFILL_MIN=16
run RT_QUAD_REFILL=1 RT_REFILL_MIN=24
run RT_QUAD_REFILL=1 RT_REFILL_MIN=8
run RT_QUAD_REFILL=1 RT_GRID_PCT=100
for V in RT_QUAD_REFILL=0 RT_QUAD_REFILL=1 "RT_QUAD_REFILL=1 RT_REFILL_MIN=16"; do
  echo "=== $V" | tee -a $OUT/q.log
  env $V RT_LANES=1 timeout -k 10 200 python3 tools/prof_frames.py --frames 16 --batch 8 2>&1 | tail -1 | cut -c1-260 | tee -a $OUT/q.log
  env $V RT_LANES=1 RT_TRACE_STATS=1 timeout -k 10 300 python3 tools/prof_frames.py --frames 16 --batch 8 2>&1 | grep "trace stats" | grep -v bounce_shadow | cut -c1-700 | tee -a $OUT/q.log
done

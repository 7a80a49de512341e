#!/bin/bash
# round 4 A/B by environment: parity subset, wall ms/frame in batches of 8 (3 repeats), stage times with one launch set in flight
#   tools/r04_ab.sh <tag> "VAR=val ..." ...   ("-" = defaults);  R04_SKIP_PARITY=1 / R04_1M=1
TAG=$1; shift; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
for V in "$@"; do
  [ "$V" == "-" ] && V="A=0"
  echo "=== [$V]" | tee -a $OUT/ab.log
  if [ -z "$R04_SKIP_PARITY" ]; then
    env $V timeout -k 10 500 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_glsl_reference.py -x -q 2>&1 | tail -1 | tee -a $OUT/ab.log
  fi
  for i in 1 2 3; do env $V timeout -k 10 120 python3 tools/wall_batch.py 1 8 2>&1 | tail -1 | tee -a $OUT/ab.log; done
  env $V RT_LANES=1 timeout -k 10 200 python3 tools/prof_frames.py --frames 16 --batch 8 2>&1 | tail -1 | cut -c1-260 | tee -a $OUT/ab.log
  if [ -n "$R04_1M" ]; then env $V RT_LANES=1 timeout -k 10 200 python3 tools/prof_frames.py --frames 3 --scene 1m 2>&1 | tail -1 | cut -c1-260 | tee -a $OUT/ab.log; fi
done

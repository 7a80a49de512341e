#!/bin/bash
# round 5 A/B by environment: optional parity subset, wall ms/frame in batches of 8 (3 repeats), stage times with one launch set in flight
#   tools/r05_ab.sh <tag> "VAR=val ..." ...   ("-" = defaults);  R05_PARITY=1 runs the parity subset first; R05_1M=1 adds the 1 M-triangle scene
TAG=$1; shift; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
for V in "$@"; do
  [ "$V" == "-" ] && V="A=0"
  echo "=== [$V]" | tee -a $OUT/ab.log
  if [ -n "$R05_PARITY" ]; then
    env $V timeout -k 10 700 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_glsl_reference.py -x -q 2>&1 | tail -3 | tee -a $OUT/ab.log || exit 1
  fi
  for i in 1 2 3; do env $V timeout -k 10 120 python3 tools/wall_batch.py 1 8 2>&1 | tail -1 | tee -a $OUT/ab.log; done
  env $V RT_LANES=1 timeout -k 10 200 python3 tools/prof_frames.py --frames 16 --batch 8 2>&1 | tail -1 | cut -c1-300 | tee -a $OUT/ab.log
  if [ -n "$R05_1M" ]; then env $V RT_LANES=1 timeout -k 10 200 python3 tools/prof_frames.py --frames 3 --scene 1m 2>&1 | tail -1 | cut -c1-300 | tee -a $OUT/ab.log; fi
done

#!/bin/bash
# A/B of traversal-kernel variants on the GPU box: parity subset + wall ms/frame (3 lanes) + serial stage times, per variant.
#   tools/ab.sh <tag> "VAR=val VAR2=val" "VAR=val" ...     ("-" = defaults)
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for V in "$@"; do
  [ "$V" == "-" ] && V=""
  echo "=== variant: [${V}]" | tee -a $OUT/ab.log
  env $V timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_glsl_reference.py -x -q 2>&1 | tail -1 | tee -a $OUT/ab.log
  env $V python3 tools/wall.py 1 2>&1 | tail -1 | tee -a $OUT/ab.log
  env $V python3 tools/wall.py 8 2>&1 | tail -1 | tee -a $OUT/ab.log
  env $V RT_LANES=1 python3 tools/prof_frames.py --frames 8 2>&1 | tail -1 | cut -c1-330 | tee -a $OUT/ab.log
  env $V RT_LANES=1 python3 tools/prof_frames.py --frames 3 --scene 1m 2>&1 | tail -1 | cut -c1-330 | tee -a $OUT/ab.log
done

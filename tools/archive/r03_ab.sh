#!/bin/bash
# A/B of kernel variants by environment: parity subset once per variant, wall ms/frame batched (bunny) + stage times (bunny batched, 1M)
#   tools/r03_ab.sh <tag> "VAR=val ..." "VAR=val" ...   ("-" = defaults)
TAG=$1; shift; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
for V in "$@"; do
  [ "$V" == "-" ] && V="A=0"
  echo "=== [$V]" | tee -a $OUT/ab.log
  env $V timeout -k 10 400 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_glsl_reference.py -x -q 2>&1 | tail -1 | tee -a $OUT/ab.log
  env $V python3 tools/wall_batch.py 1 8 2>&1 | tail -1 | tee -a $OUT/ab.log
  env $V RT_LANES=1 timeout -k 10 200 python3 tools/prof_frames.py --frames 16 --batch 8 2>&1 | tail -1 | cut -c1-260 | tee -a $OUT/ab.log
  env $V RT_LANES=1 timeout -k 10 200 python3 tools/prof_frames.py --frames 3 --scene 1m 2>&1 | tail -1 | cut -c1-260 | tee -a $OUT/ab.log
done

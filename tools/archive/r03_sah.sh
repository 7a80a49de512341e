#!/bin/bash
# VERDICT r02 item 4 measured on the GPU: any-hit launch with the 4-wide collapse of the median-split tree (default) vs a binned-SAH 4-wide tree over the same leaves
TAG=${1:-r03sah}; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
for V in A=0 RT_ANYHIT_TREE=sah; do
  echo "=== [$V]" | tee -a $OUT/sah.log
  env $V timeout -k 10 400 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_glsl_reference.py -x -q 2>&1 | tail -1 | tee -a $OUT/sah.log
  env $V python3 tools/wall_batch.py 1 8 2>&1 | tail -1 | tee -a $OUT/sah.log
  env $V RT_LANES=1 timeout -k 10 200 python3 tools/prof_frames.py --frames 16 --batch 8 2>&1 | tail -1 | cut -c1-260 | tee -a $OUT/sah.log
  env $V RT_LANES=1 RT_TRACE_STATS=1 timeout -k 10 300 python3 tools/prof_frames.py --frames 16 --batch 8 2>&1 | grep "trace stats\] shadow" | cut -c1-420 | tee -a $OUT/sah.log
  env $V RT_LANES=1 timeout -k 10 200 python3 tools/prof_frames.py --frames 3 --scene 1m 2>&1 | tail -1 | cut -c1-260 | tee -a $OUT/sah.log
  env $V RT_LANES=1 RT_TRACE_STATS=1 timeout -k 10 300 python3 tools/prof_frames.py --frames 3 --scene 1m 2>&1 | grep "trace stats\] shadow" | cut -c1-420 | tee -a $OUT/sah.log
done

#!/bin/bash
# parity subset + traversal statistics of both scenes (batched bunny, 1M frame by frame).  gpurun_out/<tag>/
TAG=${1:-r03s}; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_glsl_reference.py tests/test_gpu_baseline_configs.py -x -q 2>&1 | tail -2
python3 tools/wall_batch.py 1 8 | tail -1
RT_LANES=1 RT_TRACE_STATS=1 timeout -k 10 300 python3 tools/prof_frames.py --frames 16 --batch 8 > $OUT/stats_bunny.log 2>&1; tail -6 $OUT/stats_bunny.log | cut -c1-900
RT_LANES=1 RT_TRACE_STATS=1 timeout -k 10 300 python3 tools/prof_frames.py --frames 3 --scene 1m > $OUT/stats_1m.log 2>&1; tail -6 $OUT/stats_1m.log | cut -c1-900
RT_LANES=1 timeout -k 10 300 python3 tools/prof_frames.py --frames 3 --scene 1m 2>&1 | tail -1 | cut -c1-400

#!/bin/bash
# round 5: the parity suites under the lane / arena / queue-budget settings of tools/r04_stress.sh, re-run because the arena code changed (shadow queue 2 predicted,
# one-time shrink, overflow kernel): tools/r05_stress2.sh <tag> "<indices>"
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/${1:-r05stress2}; mkdir -p $OUT
SUITE="tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_glsl_reference.py tests/test_gpu_multirank.py tests/test_hybrid_extension.py tests/test_gpu_baseline_configs.py"
SETS=("RT_LANES=1" "RT_LANES=1 RT_QUEUE_BUDGET_MB=4" "RT_LANES=3 RT_ARENAS=1" "RT_LANES=8 RT_ARENAS=3" "RT_QUEUE_BUDGET_MB=4" "RT_QUEUE_BUDGET_MB=1 RT_LANES=2" "RT_CHUNKS_FROM_SLOTS=1 RT_QUEUE_BUDGET_MB=2"
      "RT_Q2_CAP=256" "RT_Q2_CAP=4096 RT_LANES=1 RT_QUEUE_BUDGET_MB=8" "RT_Q2_PREDICT=0 RT_ARENAS=4" "RT_Q2_CAP=64 RT_ARENAS=1 RT_LANES=3 RT_QNODES=2")
for i in ${2:-0 1 2 3 4 5 6 7 8 9 10}; do
  V=${SETS[$i]}
  echo -n "[$V] " | tee -a $OUT/stress.log
  env $V timeout -k 10 700 python3 -m pytest $SUITE -x -q -m gpu 2>&1 | tail -1 | tee -a $OUT/stress.log
done

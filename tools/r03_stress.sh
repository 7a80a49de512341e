#!/bin/bash
# robustness: the parity suites under unusual lane counts and a tiny ray-queue budget (every frame cut into many chunks, hit count read back)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03stress
for V in "RT_LANES=1" "RT_LANES=1 RT_QUEUE_BUDGET_MB=4" "RT_LANES=3" "RT_LANES=8" "RT_QUEUE_BUDGET_MB=4" "RT_QUEUE_BUDGET_MB=1 RT_LANES=2" "RT_GRID_PCT=30" "RT_QUAD_REFILL=1 RT_REFILL_MIN=8" "RT_ANYHIT_TREE=sah RT_CHUNK=64" "RT_CHUNK=8 RT_CHUNK_PRIMARY=8 RT_MIN_SEARCH=64" "RT_CHUNKS_FROM_SLOTS=1 RT_QUEUE_BUDGET_MB=2"; do
  echo -n "[$V] " | tee -a gpurun_out/r03stress/stress.log
  env $V timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_glsl_reference.py tests/test_gpu_multirank.py tests/test_hybrid_extension.py tests/test_gpu_baseline_configs.py -x -q -m gpu 2>&1 | tail -1 | tee -a gpurun_out/r03stress/stress.log
done

#!/bin/bash
# round 4 robustness: the parity suites under unusual settings (round 3's eleven + this round's options and arena / estimate knobs), then the fuzz sweeps at 100x
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/${1:-r04stress}; mkdir -p $OUT
for V in "RT_LANES=1" "RT_LANES=1 RT_QUEUE_BUDGET_MB=4" "RT_LANES=3 RT_ARENAS=1" "RT_LANES=8 RT_ARENAS=3" "RT_QUEUE_BUDGET_MB=4" "RT_QUEUE_BUDGET_MB=1 RT_LANES=2" "RT_GRID_PCT=30" "RT_QUAD_REFILL=1 RT_REFILL_MIN=8" \
         "RT_ANYHIT_TREE=sah RT_CHUNK=64" "RT_CHUNK=8 RT_CHUNK_PRIMARY=8 RT_MIN_SEARCH=64" "RT_CHUNKS_FROM_SLOTS=1 RT_QUEUE_BUDGET_MB=2" \
         "RT_COOP=1 RT_PACKET_AO=1" "RT_BIN_GI=1 RT_NEAR_FIRST=1 RT_LEAFB=4" "RT_CU_SPLIT=2 RT_PACKET_AO=1 RT_QUEUE_BUDGET_MB=3" "RT_HYBRID_RATIO_Q=0.01 RT_HYBRID_RATIO_L=0.01 RT_ARENAS=4" \
         "RT_QNODES=2 RT_LANES=1" "RT_QNODES=2 RT_QUAD_REFILL=1 RT_CHUNK=8 RT_ANYHIT_TREE=sah" "RT_QNODES=1 RT_GRID_PCT=30 RT_QUEUE_BUDGET_MB=1"; do
  echo -n "[$V] " | tee -a $OUT/stress.log
  env $V timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_glsl_reference.py tests/test_gpu_multirank.py tests/test_hybrid_extension.py tests/test_gpu_baseline_configs.py -x -q -m gpu 2>&1 | tail -1 | tee -a $OUT/stress.log
done
RT_FUZZ_CASES=1500 timeout -k 10 1100 python3 -m pytest tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -1 | tee -a $OUT/stress.log
RT_FUZZ_CASES=300 timeout -k 10 1100 python3 -m pytest tests/test_hybrid_extension.py -x -q -m gpu -k random 2>&1 | tail -1 | tee -a $OUT/stress.log

#!/bin/bash
# round 5 robustness: the parity suites with this round's record forms forced on (alone and together with unusual scheduler / queue settings), the hybrid extension under
# its bounds check, then the fuzz sweeps at 100x with each form.   tools/r05_stress.sh <tag> [part]   part 1: settings matrix, part 2: fuzz sweeps
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/${1:-r05stress}; mkdir -p $OUT; PART=${2:-12}
SUITE="tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_glsl_reference.py tests/test_gpu_multirank.py tests/test_hybrid_extension.py tests/test_gpu_baseline_configs.py"
SETS=( "RT_FUSED=1" "RT_IMPLICIT=1" "RT_FUSED=1 RT_LANES=1 RT_QUEUE_BUDGET_MB=4" "RT_IMPLICIT=1 RT_CHUNK=8 RT_CHUNK_PRIMARY=8 RT_MIN_SEARCH=64" "RT_FUSED=1 RT_QUAD_REFILL=1 RT_REFILL_MIN=8 RT_GRID_PCT=30" \
         "RT_IMPLICIT=1 RT_LANES=8 RT_ARENAS=3 RT_QNODES=2" "RT_HYBRID_CHECK=1 RT_HYBRID_RATIO_Q=0.01 RT_HYBRID_RATIO_L=0.01 RT_QUEUE_BUDGET_MB=2" "RT_HYBRID_WAVES=5 RT_FUSED=1" "RT_HYBRID_WAVES=3 RT_IMPLICIT=1")
if [[ $PART == *1* ]]; then
for i in ${R05_PICK:-0 1 2 3 4 5 6 7 8}; do
  V=${SETS[$i]}
  echo -n "[$V] " | tee -a $OUT/stress.log
  env $V timeout -k 10 700 python3 -m pytest $SUITE -x -q -m gpu 2>&1 | tail -1 | tee -a $OUT/stress.log
done
fi
if [[ $PART == *2* ]]; then
for V in "RT_FUSED=1" "RT_IMPLICIT=1"; do
  echo -n "[fuzz x100 $V] " | tee -a $OUT/stress.log
  env $V RT_FUZZ_CASES=1500 timeout -k 10 1100 python3 -m pytest tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -1 | tee -a $OUT/stress.log
done
echo -n "[hybrid fuzz x100, RT_HYBRID_CHECK=1] " | tee -a $OUT/stress.log
RT_HYBRID_CHECK=1 RT_FUZZ_CASES=300 timeout -k 10 1100 python3 -m pytest tests/test_hybrid_extension.py -x -q -m gpu -k random 2>&1 | tail -1 | tee -a $OUT/stress.log
fi

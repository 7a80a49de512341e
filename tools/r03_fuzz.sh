#!/bin/bash
# the randomised parity sweeps at 100x their default size (BVH scenes on both pipelines, analytic scene, hybrid extension on both pipelines)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03fuzz
RT_FUZZ_CASES=${1:-1500} timeout -k 10 1100 python3 -m pytest tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -2 | tee gpurun_out/r03fuzz/fuzz.log
RT_FUZZ_CASES=${2:-300} timeout -k 10 1100 python3 -m pytest tests/test_hybrid_extension.py -x -q -m gpu -k random 2>&1 | tail -2 | tee -a gpurun_out/r03fuzz/fuzz.log

#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03y
timeout -k 10 600 python3 -m pytest tests/test_hybrid_extension.py -x -q -m gpu 2>&1 | tail -1
timeout -k 10 600 python3 bench.py > gpurun_out/r03y/bench.json 2> gpurun_out/r03y/bench.err; echo "bench rc=$?"

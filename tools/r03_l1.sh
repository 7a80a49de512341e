#!/bin/bash
cd $GRAFT_REPO_ROOT
RT_LANES=1 timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_multirank.py tests/test_gpu_baseline_configs.py -x -q -m gpu 2>&1 | tail -40
RT_LANES=1 python3 tools/prof_frames.py --frames 16 --batch 8 2>&1 | tail -1 | cut -c1-240

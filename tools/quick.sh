#!/bin/bash
# quick GPU check: parity tests + wall time at world 1 and 8 + serial stage breakdown
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_multirank.py -x -q 2>&1 | tail -2
python3 tools/wall.py 1 && python3 tools/wall.py 8 && RT_LANES=1 python3 tools/prof_frames.py --frames 6 2>&1 | tail -1 | cut -c1-300

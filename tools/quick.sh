#!/bin/bash
# quick GPU check: a few parity tests + wall time at world 1 and 8 + serial stage breakdown
python -m pytest tests/test_gpu_parity.py -x -q -k "wavefront or full_size or chunked or traversal" 2>&1 | tail -2
python3 tools/wall.py 1; python3 tools/wall.py 8
RT_LANES=1 python3 tools/prof_frames.py --frames 6 2>&1 | tail -3 | cut -c1-600

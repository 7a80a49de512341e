#!/bin/bash
for r in 32 24 16 8 4 1; do echo -n "refillMin=$r: "; RT_REFILL_MIN=$r python3 tools/wall.py 1 | tail -1; RT_REFILL_MIN=$r python3 tools/wall.py 8 | tail -1; RT_REFILL_MIN=$r RT_LANES=1 python3 tools/prof_frames.py --frames 6 2>&1 | tail -1 | cut -c1-200; done

#!/bin/bash
for cfg in "32 16" "24 16" "16 16" "16 8" "8 8" "24 8" "32 24"; do set -- $cfg; echo -n "refillMin=$1 minSearch=$2: "; RT_REFILL_MIN=$1 RT_MIN_SEARCH=$2 python3 tools/wall.py 1 2>/dev/null | tail -1; RT_REFILL_MIN=$1 RT_MIN_SEARCH=$2 RT_LANES=1 python3 tools/prof_frames.py --frames 6 2>/dev/null | tail -1 | cut -c1-200; done

#!/bin/bash
for w in 8 4; do for cfg in "100 4" "50 4" "25 4" "50 6" "25 8" "12 8" "50 3" "25 6"; do set -- $cfg; echo -n "world=$w grid=$1% lanes=$2: "; RT_GRID_PCT=$1 RT_LANES=$2 python3 tools/wall.py $w 2>/dev/null | tail -1; done; done

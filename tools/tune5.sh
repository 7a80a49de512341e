#!/bin/bash
for w in 8 4 2; do for l in 2 3 4 5 6 8; do echo -n "world=$w lanes=$l: "; RT_LANES=$l python3 tools/wall.py $w 2>/dev/null | tail -1; done; done

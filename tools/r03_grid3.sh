#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { echo -n "[$*] "; for i in 1 2 3; do env "$@" timeout -k 10 120 python3 tools/wall_batch.py 1 8 2>&1 | tail -1 | awk '{printf "%s ", $5}'; done; echo; }
run A=0
run RT_GRID_PCT=100
run RT_GRID_PCT=85
run RT_GRID_PCT=67
run RT_CHUNK=512
run RT_CHUNK=256
run RT_LANES=3
run RT_LANES=5
run A=0

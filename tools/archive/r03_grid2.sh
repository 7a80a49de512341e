#!/bin/bash
TAG=${1:-r03g2}; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
run() { echo -n "[$*] " | tee -a $OUT/grid.log; for i in 1 2 3; do env "$@" timeout -k 10 120 python3 tools/wall_batch.py 1 8 2>&1 | tail -1 | awk '{printf "%s ", $5}' | tee -a $OUT/grid.log; done; echo | tee -a $OUT/grid.log; }
run A=0
run RT_GRID_PCT=75 RT_CHUNK=384
run RT_GRID_PCT=75 RT_CHUNK=384 RT_LANES=4
run RT_GRID_PCT=67 RT_CHUNK=384
run RT_GRID_PCT=67 RT_CHUNK=384 RT_LANES=4
run RT_GRID_PCT=75 RT_CHUNK=512 RT_LANES=4
run RT_GRID_PCT=80 RT_CHUNK=384 RT_GRID_PCT_PRIMARY=100
run RT_CHUNK=384
run A=0

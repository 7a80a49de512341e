#!/bin/bash
# persistent-grid size x frame lanes x run length in the as-benchmarked mode (batches of 8): wall ms/frame
TAG=${1:-r03g}; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
run() { echo -n "[$*] " | tee -a $OUT/grid.log; env "$@" timeout -k 10 120 python3 tools/wall_batch.py 1 8 2>&1 | tail -1 | tee -a $OUT/grid.log; }
run A=0
for g in 50 60 67 75 85; do run RT_GRID_PCT=$g; run RT_GRID_PCT=$g RT_GRID_PCT_PRIMARY=100; done
for g in 60 67 75; do run RT_GRID_PCT=$g RT_LANES=4; run RT_GRID_PCT=$g RT_LANES=2; run RT_GRID_PCT=$g RT_CHUNK=512; run RT_GRID_PCT=$g RT_CHUNK=384;  done
run RT_GRID_PCT=67 RT_LANES=4 RT_CHUNK=512
run RT_GRID_PCT=75 RT_LANES=4 RT_CHUNK=512
run A=0

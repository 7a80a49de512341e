#!/bin/bash
# knob sweep in the as-benchmarked mode (batches of 8 frames, 3 lanes): wall ms/frame per setting
TAG=${1:-r03k}; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
run() { echo -n "[$*] " | tee -a $OUT/knobs.log; env "$@" timeout -k 10 120 python3 tools/wall_batch.py 1 8 2>&1 | tail -1 | tee -a $OUT/knobs.log; }
run A=0
run A=0
for v in 2 4 5; do run RT_LANES=$v; done
for v in 16 24 40 48; do run RT_REFILL_MIN=$v; done
for v in 4 8 24 32; do run RT_MIN_SEARCH=$v; done
for v in 64 128 256 512; do run RT_CHUNK=$v; done
for v in 32 128; do run RT_CHUNK_PRIMARY=$v; done
run RT_LEAFB=4
for v in 75 125 150; do run RT_GRID_PCT=$v; done
echo -n "[K=16] "; timeout -k 10 120 python3 tools/wall_batch.py 1 16 | tail -1
echo -n "[K=4] "; timeout -k 10 120 python3 tools/wall_batch.py 1 4 | tail -1

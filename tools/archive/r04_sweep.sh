#!/bin/bash
# round 4: grid size / lanes / arenas sweep with the seven-wave any-hit kernels (wall ms per frame, batches of 8, three repeats)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-r04sw}; mkdir -p $OUT; cd $R
run() { echo -n "[$*] " | tee -a $OUT/sweep.log; for i in 1 2 3; do env "$@" timeout -k 10 120 python3 tools/wall_batch.py 1 8 2>&1 | tail -1 | awk '{printf "%s ", $5}' | tee -a $OUT/sweep.log; done; echo | tee -a $OUT/sweep.log; }
run A=0
run RT_GRID_PCT=60
run RT_GRID_PCT=67
run RT_GRID_PCT=85
run RT_GRID_PCT=100
run RT_LANES=3
run RT_LANES=5
run RT_LANES=6
run RT_LANES=6 RT_ARENAS=3
run RT_CHUNK=256
run RT_CHUNK=512
run A=0

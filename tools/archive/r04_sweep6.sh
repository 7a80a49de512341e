#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-r04sw6}; mkdir -p $OUT; cd $R
run() { echo -n "[$*] " | tee -a $OUT/sweep.log; for i in 1 2 3; do env "$@" timeout -k 10 120 python3 tools/wall_batch.py ${WORLD:-1} ${KK:-8} 2>&1 | tail -1 | awk '{printf "%s ", $5}' | tee -a $OUT/sweep.log; done; echo | tee -a $OUT/sweep.log; }
run A=0
run RT_GRID_PCT_PRIMARY=75
run RT_GRID_PCT_PRIMARY=50
run A=0
run RT_LANES=3
run RT_LANES=3 RT_ARENAS=3
run RT_ARENAS=3
run RT_ARENAS=4
run A=0
KK=12 run A=0
KK=16 run A=0
KK=6 run A=0

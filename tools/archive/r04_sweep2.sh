#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-r04sw2}; mkdir -p $OUT; cd $R
run() { echo -n "[$*] " | tee -a $OUT/sweep.log; for i in 1 2 3; do env "$@" timeout -k 10 120 python3 tools/wall_batch.py 1 8 2>&1 | tail -1 | awk '{printf "%s ", $5}' | tee -a $OUT/sweep.log; done; echo | tee -a $OUT/sweep.log; }
run A=0
run RT_CHUNK=512
run A=0
run RT_CHUNK=768
run A=0
run RT_CHUNK=1024
run A=0
run RT_CHUNK=512
run RT_CHUNK=512 RT_GRID_PCT=67
run A=0

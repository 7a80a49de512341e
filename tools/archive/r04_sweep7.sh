#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-r04sw7}; mkdir -p $OUT; cd $R
run() { echo -n "[$*] " | tee -a $OUT/sweep.log; for i in 1 2 3; do env "$@" timeout -k 10 120 python3 tools/wall_batch.py ${WORLD:-1} ${KK:-8} 2>&1 | tail -1 | awk '{printf "%s ", $5}' | tee -a $OUT/sweep.log; done; echo | tee -a $OUT/sweep.log; }
run A=0
run RT_REFILL_MIN=24
run RT_REFILL_MIN=16
run A=0
run RT_REFILL_MIN=28
run RT_REFILL_MIN=24 RT_MIN_SEARCH=12
run RT_REFILL_MIN=40
run A=0

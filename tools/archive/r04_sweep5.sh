#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-r04sw5}; mkdir -p $OUT; cd $R
run() { echo -n "[$*] " | tee -a $OUT/sweep.log; for i in 1 2 3; do env "$@" timeout -k 10 120 python3 tools/wall_batch.py ${WORLD:-1} ${KK:-8} 2>&1 | tail -1 | awk '{printf "%s ", $5}' | tee -a $OUT/sweep.log; done; echo | tee -a $OUT/sweep.log; }
run A=0
run RT_CHUNK_PRIMARY=256
run RT_CHUNK_PRIMARY=384
run RT_CHUNK_PRIMARY=512
run A=0
run RT_CHUNK_PRIMARY=768
run RT_CHUNK_PRIMARY=1024
run A=0
echo "--- frame by frame (K = 1)" | tee -a $OUT/sweep.log
KK=1 run A=0
KK=1 run RT_CHUNK_PRIMARY=128
KK=1 run RT_CHUNK_PRIMARY=256
echo "--- one rank of eight, batches of 8" | tee -a $OUT/sweep.log
WORLD=8 run A=0
WORLD=8 run RT_CHUNK_PRIMARY=128
WORLD=8 run RT_CHUNK_PRIMARY=256

#!/bin/bash
# round 5: the driver's invocation (--steps 20 --warmup 5) at different batch lengths: does the 34 ms timed region fill the four frame lanes better with shorter batches?
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-r05sr}; mkdir -p $OUT; cd $R
for B in 8 5 4 10 7; do
  for i in 1 2 3; do
    timeout -k 10 200 python3 bench.py --gpus 1 --steps 20 --warmup 5 --batch $B --cpu-seconds 0 --no-default-camera --no-frame-by-frame --no-diagnostics --no-run-b --parity-window 0 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('batch $B ms_per_step', round(d['ms_per_step'],4))" | tee -a $OUT/short.log
  done
done

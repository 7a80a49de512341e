#!/bin/bash
# round 5: the driver's own invocation, repeated, per environment setting (ms per step of a 34 ms timed region scatter by +-2 %)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-r05dl}; shift; mkdir -p $OUT; cd $R
for V in "$@"; do
  [ "$V" == "-" ] && V="A=0"
  for i in 1 2 3 4; do
    env $V timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-seconds 0 --no-run-b --parity-window 0 --no-default-camera --no-diagnostics 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$V] ms_per_step', round(d['ms_per_step'],4), 'in_use_GB', round(d['config']['hbm']['in_use_GB'],1))" | tee -a $OUT/driver_like.log
  done
done

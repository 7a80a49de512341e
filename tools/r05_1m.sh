#!/bin/bash
# round 5: the 1 M-triangle scene (configs[4], 4 spp): stage times with one launch set in flight + bench.py --scene 1m ms per step, per environment setting
TAG=$1; shift; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
for V in "$@"; do
  [ "$V" == "-" ] && V="A=0"
  echo "=== [$V]" | tee -a $OUT/1m.log
  env $V RT_LANES=1 timeout -k 10 200 python3 tools/prof_frames.py --frames 8 --batch 4 --scene 1m 2>&1 | tail -1 | cut -c1-330 | tee -a $OUT/1m.log
  env $V timeout -k 10 300 python3 bench.py --scene 1m --steps 16 --warmup 8 --cpu-seconds 0 --no-default-camera --no-frame-by-frame --no-diagnostics --no-run-b --parity-window 0 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench --scene 1m ms_per_step', round(d['ms_per_step'],2))" | tee -a $OUT/1m.log
done

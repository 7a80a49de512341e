#!/bin/bash
# round 4: the driver's short run (--steps 20 --warmup 5): how the 20 frames are best dealt over batches (pipeline fill / drain matter at 35 ms)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-r04short}; mkdir -p $OUT; cd $R
for B in 8 7 5 4 10; do
  for i in 1 2 3; do
    timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --batch $B --cpu-seconds 0 --no-default-camera --no-diagnostics --parity-window 0 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('batch $B  ms/step %.4f  (frame by frame %.4f)' % (d['ms_per_step'], d['config']['ms_per_step_frame_by_frame']))" | tee -a $OUT/short.log
  done
done

#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python3 bench.py --gpus 1 --launch --force-gather --gather torch --steps 12 --warmup 4 --cpu-seconds 0 --no-default-camera 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('torch gather path: ms/step %.3f same %s gather %s' % (d['ms_per_step'], d['config']['batched_equals_frame_by_frame'], d['config']['gather']['path']))"
timeout -k 10 600 python3 bench.py --gpus 1 --launch --force-gather --steps 12 --warmup 4 --cpu-seconds 0 --no-default-camera --gather-every 32 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('native gather, every 32: ms/step %.3f same %s gather %s' % (d['ms_per_step'], d['config']['batched_equals_frame_by_frame'], d['config']['gather']['path']))"

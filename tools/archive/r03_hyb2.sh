#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python3 bench.py --hybrid --spp 16 --gi-bounces 4 --steps 6 --warmup 2 --cpu-seconds 0 --no-default-camera --no-frame-by-frame 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('16spp/4b ms/frame %.2f' % d['ms_per_step'], {k: round(v, 2) for k, v in d['stage_ms_per_frame'].items()})"
timeout -k 10 600 python3 bench.py --hybrid --spp 4 --gi-bounces 1 --steps 6 --warmup 2 --cpu-seconds 0 --no-default-camera --no-frame-by-frame 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('4spp/1b ms/frame %.2f' % d['ms_per_step'])"

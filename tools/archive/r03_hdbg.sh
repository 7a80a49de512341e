#!/bin/bash
cd $GRAFT_REPO_ROOT
RT_HYBRID_DEBUG=1 RT_LANES=1 timeout -k 10 300 python3 bench.py --hybrid --spp 16 --gi-bounces 4 --steps 1 --warmup 0 --cpu-seconds 0 --no-default-camera --no-frame-by-frame 2>&1 | grep "\[hybrid\]" | head -40

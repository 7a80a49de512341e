#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { echo -n "[$*] "; for i in 1 2 3; do env "$@" timeout -k 10 120 python3 tools/wall_batch.py 1 8 2>&1 | tail -1 | awk '{printf "%s ", $5}'; done; echo; }
run A=0
run RT_QUEUE_BUDGET_MB=16384
run RT_QUEUE_BUDGET_MB=4096
echo -n "[K=16 budget 32G] "; RT_QUEUE_BUDGET_MB=32768 python3 tools/wall_batch.py 1 16 | tail -1
echo -n "[K=12 budget 16G] "; RT_QUEUE_BUDGET_MB=16384 python3 tools/wall_batch.py 1 12 | tail -1

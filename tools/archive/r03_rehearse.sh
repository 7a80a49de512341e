#!/bin/bash
# N rank processes on the one GPU of the box (gloo, host-staged gathers): bench.py's multi-process path end to end
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python3 -m pytest tests/test_bench_contract.py -x -q -k "ranks_as_processes or self_launch" 2>&1 | tail -15
timeout -k 10 600 python3 bench.py --gpus 2 --rehearse-one-gpu --steps 16 --warmup 4 --cpu-seconds 0 2>gpurun_out/rehearse.err | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('2 processes on one GPU: ms/step %.3f n_gpus %d same %s assembled %s | %s' % (d['ms_per_step'], d['n_gpus'], d['config']['batched_equals_frame_by_frame'], d['config']['assembled_equals_single_rank'], d['config']['gather']['path']))"
tail -5 gpurun_out/rehearse.err

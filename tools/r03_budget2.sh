#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03bud
timeout -k 10 900 python3 -m pytest tests/test_gpu_baseline_configs.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2
timeout -k 10 600 python3 bench.py > gpurun_out/r03bud/bench.json 2> gpurun_out/r03bud/bench.err; echo "bench rc=$?"; python3 -c "
import json; d=json.loads(open('gpurun_out/r03bud/bench.json').read().strip().splitlines()[-1]); r=d['roofline']
print('ms/step %.4f  fbf %.3f  same %s  dominant %s %.3f ms x %d launches (%.1f frames each)  hbm frac %.4f  l1 frac %.3f  default cam %.4f' % (d['ms_per_step'], d['config']['ms_per_step_frame_by_frame'], d['config']['batched_equals_frame_by_frame'], r['kernel'], r['avg_launch_ms'], r['launches'], r['frames_per_launch'], r['frac'], r['l1_gather']['frac'], d['default_camera']['ms_per_step']))"
bash tools/other_configs.sh r03bud 2>&1 | grep -v "^$"

#!/bin/bash
# the driver's invocations, as the round-end harness issues them
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03drv
SECONDS=0; python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03drv/n1.json 2> gpurun_out/r03drv/n1.err; echo "rc=$? wall ${SECONDS}s"
python3 -c "
import json; d=json.loads(open('gpurun_out/r03drv/n1.json').read().strip().splitlines()[-1]); r=d['roofline']
print('ms/step %.4f value %.0f same %s roofline %s frac %.4f l1 %.3f traffic %s cpu %.2f/%.2f' % (d['ms_per_step'], d['value'], d['config']['batched_equals_frame_by_frame'], r['kernel'], r['frac'], r['l1_gather']['frac'], r['traffic'], d['cpu_baseline']['value'], d['cpu_baseline']['single_thread']['value']))"
python3 -c "import __graft_entry__ as g; g.smoke()"

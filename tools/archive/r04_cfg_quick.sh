#!/bin/bash
# quick check of the multi-chunk configurations + headline after an arena policy change
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-r04cq}; mkdir -p $OUT; cd $R
timeout -k 10 600 python3 -m pytest tests/test_gpu_baseline_configs.py tests/test_gpu_parity.py -x -q 2>&1 | tail -1
run() { n=$1; shift; timeout -k 10 600 python3 bench.py --steps 24 --warmup 8 --cpu-seconds 0 --no-default-camera --no-diagnostics "$@" > $OUT/cfg_$n.json 2> $OUT/cfg_$n.err; python3 - <<PY
import json
d=json.loads(open("$OUT/cfg_$n.json").read().strip().splitlines()[-1])
print("$n ms/frame %.3f  hbm in use %.1f GB arenas %.1f GB (%d)  parity %s" % (d["ms_per_step"], d["config"]["hbm"]["in_use_GB"], d["config"]["hbm"]["ray_queue_arenas_GB"], d["config"]["hbm"]["ray_queue_arenas"], d["parity"]["ok"]))
PY
}
run head
run spp16 --spp 16
run 1m4 --scene 1m --spp 4
run 4k16 --size 3840x2160 --spp 16
run 1m64 --scene 1m --spp 64 --steps 8 --warmup 8

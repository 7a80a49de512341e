#!/bin/bash
# round 4: the whole GPU suite + the default bench line (what the driver runs at round end)
TAG=${1:-r04full}; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
timeout -k 10 1000 python3 -m pytest tests -x -q -m gpu > $OUT/pytest.log 2>&1; rc=$?; echo "pytest rc $rc"; tail -4 $OUT/pytest.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 400 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc $?"
timeout -k 10 400 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_like.json 2> $OUT/bench_driver_like.err; echo "bench (driver-like) rc $?"
python3 - <<PY
import json
for f in ("bench", "bench_driver_like"):
    d=json.load(open("$OUT/%s.json" % f))
    print(f, "ms/step %.4f" % d["ms_per_step"], "value %.0f" % d["value"], "parity", d["parity"]["ok"], d["parity"]["bit_diff"], "roofline", d["roofline"]["kernel"], "%.4f" % d["roofline"]["frac"], "l1 %.3f" % d["roofline"]["l1_gather"]["frac"], d["config"]["hbm"]["in_use_GB"], d["config"]["hbm"]["ray_queue_arenas_GB"])
PY

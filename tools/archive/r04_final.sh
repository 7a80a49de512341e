#!/bin/bash
# round 4 end-of-round sequence: GPU suite, PMC passes that stamp profiles/r04_traffic_*.json, then the bench lines with the fresh stamp
TAG=${1:-r04fin}; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
timeout -k 10 1000 python3 -m pytest tests -x -q -m gpu > $OUT/pytest.log 2>&1; rc=$?; echo "pytest rc $rc"; tail -2 $OUT/pytest.log
[ $rc -ne 0 ] && exit $rc
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash tools/r03_profile.sh $TAG trace pmc_bunny pmc_1m > $OUT/profile.log 2>&1; tail -2 $OUT/profile.log
python3 tools/r04_summarize.py $TAG > /dev/null 2>&1     # copies the traffic stamps into profiles/ (of this scratch copy) so that the bench lines below carry them
timeout -k 10 400 python3 bench.py > $OUT/bench_line.json 2> $OUT/bench_line.err; echo "bench rc $?"
timeout -k 10 400 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_like.json 2> $OUT/bench_driver_like.err; echo "bench (driver-like) rc $?"
timeout -k 10 300 python3 bench.py --scene 1m --steps 20 --warmup 3 --cpu-seconds 0 --no-default-camera > $OUT/bench_line_1m.json 2>/dev/null; echo "bench 1m rc $?"
python3 - <<PY
import json
for f in ("bench_line", "bench_driver_like", "bench_line_1m"):
    d=json.load(open("$OUT/%s.json" % f)); r=d["roofline"]
    print(f, "ms/step %.4f" % d["ms_per_step"], "parity", d["parity"]["ok"], d["parity"]["bit_diff"], "roofline", r["kernel"], "%.4f" % r["frac"], "traffic", r["traffic"], (r["traffic_source"] or {}).get("kind"), "l1", r["l1_gather"] and round(r["l1_gather"]["frac"], 3), "hbm", round(d["config"]["hbm"]["in_use_GB"], 1))
PY

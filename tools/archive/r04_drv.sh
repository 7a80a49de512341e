#!/bin/bash
# Driver-like and default bench lines, several times within one box:  tools/r04_drv.sh <tag>
TAG=${1:-r04drv}; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
python -m pytest tests/test_bench_contract.py -x -q -m gpu > $OUT/contract.log 2>&1; tail -2 $OUT/contract.log
for i in 1 2 3; do
  python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/drv$i.json 2> $OUT/drv$i.err && python - <<PY
import json; d=json.load(open("$OUT/drv$i.json")); print("drv$i", d["ms_per_step"], d["value"], d["parity"]["rmse"], d["roofline"]["traffic"] if not isinstance(d["roofline"]["traffic"],dict) else "traffic ok", sorted(d.get("stage_ms_per_frame",{}).items())[:3])
PY
done
python bench.py > $OUT/full.json 2> $OUT/full.err && python - <<PY
import json; d=json.load(open("$OUT/full.json")); print("full", d["ms_per_step"], d["value"], d["parity"]["rmse"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"])
PY
for w in 5 40 5 40; do
  python bench.py --gpus 1 --steps 20 --warmup $w --no-run-b --cpu-seconds 0 --no-default-camera --no-frame-by-frame > $OUT/w$w.json 2> $OUT/w$w.err && python - <<PY
import json; d=json.load(open("$OUT/w$w.json")); print("warmup $w", d["ms_per_step"])
PY
done

#!/bin/bash
# round 4: the hybrid extension after its rewrite (compact queue + log arena, host out of the pass loop): tests, then bench lines
TAG=${1:-r04h}; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
timeout -k 10 900 python3 -m pytest tests/test_hybrid_extension.py -x -q > $OUT/pytest.log 2>&1; echo "pytest rc $?"; tail -4 $OUT/pytest.log
[ -n "$R04_TESTS_ONLY" ] && exit 0
run() { name=$1; shift; timeout -k 10 600 python3 bench.py --hybrid "$@" --cpu-seconds 0 --no-default-camera --no-frame-by-frame > $OUT/$name.json 2> $OUT/$name.err; echo "$name rc $?"; python3 - <<PY
import json
d=json.load(open("$OUT/$name.json"))
print("$name", "ms/frame %.2f" % d["ms_per_step"], {k: round(v, 2) for k, v in d["stage_ms_per_frame"].items()}, d["config"]["hbm"], d.get("parity", {}) and d["parity"]["ok"])
PY
}
run h1080_16_4 --spp 16 --gi-bounces 4 --steps 6 --warmup 2
run h1080_4_1 --spp 4 --gi-bounces 1 --steps 6 --warmup 2
run h4k_16_4 --size 3840x2160 --spp 16 --gi-bounces 4 --steps 3 --warmup 1

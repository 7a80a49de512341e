#!/bin/bash
TAG=${1:-r03h}; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
timeout -k 10 900 python3 -m pytest tests/test_hybrid_extension.py tests/test_gpu_baseline_configs.py::test_mixed_sequences_fall_back_per_run tests/test_cli.py -x -q -m gpu 2>&1 | tail -5
for P in auto mega; do
  echo "== pipeline $P"
  timeout -k 10 600 python3 bench.py --hybrid --spp 16 --gi-bounces 4 --steps 6 --warmup 2 --cpu-seconds 0 --no-default-camera --pipeline $P > $OUT/hybrid_$P.json 2> $OUT/hybrid_$P.err; echo "rc=$?"; tail -2 $OUT/hybrid_$P.err
  python3 - <<PY
import json
d=json.loads(open("$OUT/hybrid_$P.json").read().strip().splitlines()[-1])
print("  ms/frame %.2f  fbf %.2f  same %s  stages %s" % (d["ms_per_step"], d["config"]["ms_per_step_frame_by_frame"] or 0, d["config"]["batched_equals_frame_by_frame"], {k: round(v, 2) for k, v in (d.get("stage_ms_per_frame") or {}).items()}))
PY
done
timeout -k 10 600 python3 bench.py --hybrid --spp 4 --gi-bounces 1 --steps 6 --warmup 2 --cpu-seconds 0 --no-default-camera > $OUT/hybrid_4_1.json 2> $OUT/hybrid_4_1.err; echo "rc=$?"; python3 -c "
import json; d=json.loads(open('$OUT/hybrid_4_1.json').read().strip().splitlines()[-1]); print('  4spp/1 bounce ms/frame %.2f' % d['ms_per_step'])"

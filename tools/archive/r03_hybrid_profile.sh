#!/bin/bash
# kernel trace of the staged hybrid extension (bench.py --hybrid --spp 16 --gi-bounces 4) + its bench lines on both pipelines
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r03hp; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--hybrid --spp 16 --gi-bounces 4 --steps 6 --warmup 2 --cpu-seconds 0 --no-default-camera"
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py $ARGS --no-frame-by-frame > $OUT/trace.log 2>&1
cp $OUT/trace/*/*kernel_stats.csv $OUT/hybrid_kernel_stats.csv 2>/dev/null; head -6 $OUT/hybrid_kernel_stats.csv | cut -c1-200
cd $R
for P in auto mega; do timeout -k 10 600 python3 bench.py $ARGS --pipeline $P > $OUT/hybrid_$P.json 2> $OUT/hybrid_$P.err; echo "$P rc=$?"; done
python3 - <<PY
import json
for p in ("auto","mega"):
    d=json.loads(open("$OUT/hybrid_%s.json" % p).read().strip().splitlines()[-1])
    print(p, "ms/frame %.2f same %s" % (d["ms_per_step"], d["config"]["batched_equals_frame_by_frame"]), {k: round(v, 2) for k, v in (d.get("stage_ms_per_frame") or {}).items()})
PY

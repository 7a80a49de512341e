#!/bin/bash
# round 5: run B (1080p, 16 spp, 4 bounces) ms per step + stage split of the staged hybrid pipeline per environment setting:  tools/r05_hybrid_time.sh <tag> "VAR=val" ...
TAG=$1; shift; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
ARGS="--hybrid --spp 16 --gi-bounces 4 --steps 6 --warmup 2 --cpu-seconds 0 --no-default-camera --no-frame-by-frame --no-diagnostics --parity-window 64x32"
for V in "$@"; do
  [ "$V" == "-" ] && V="A=0"
  echo "=== [$V]" | tee -a $OUT/hybrid.log
  env $V timeout -k 10 300 python3 bench.py $ARGS 2>$OUT/err.txt | tail -1 > $OUT/line.json
  python3 - <<PY | tee -a $OUT/hybrid.log
import json
d = json.load(open("$OUT/line.json"))
c = d["config"]
print("ms_per_step", round(d["ms_per_step"], 2), "stages", {k: round(v, 2) for k, v in (c.get("stage_ms_per_frame") or {}).items()}, "parity", (d.get("parity") or {}).get("ok"), "hybrid_arena_GB", c.get("hbm", {}).get("hybrid_arena_GB"))
PY
done

#!/bin/bash
# bench.py on the other BASELINE.json configurations (one GPU): DESIGN.md section 6 table.  gpurun_out/<tag>/cfg_*.json
TAG=${1:-cfg}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
run() { n=$1; shift; timeout -k 10 600 python3 bench.py --steps 24 --warmup 8 --cpu-seconds 0 --no-default-camera "$@" > $OUT/cfg_$n.json 2> $OUT/cfg_$n.err; echo "$n rc=$?"; python3 - <<PY
import json
d=json.loads(open("$OUT/cfg_$n.json").read().strip().splitlines()[-1]); r=d["roofline"]
print("  ms/frame %.2f  Mray/s %.0f  traversed Mray/s %.0f  traversed rays/frame %d  dominant %s %.2f ms  l1 frac %s" % (d["ms_per_step"], d["value"], d["value_traversed"], d["config"]["rays_traversed_per_frame"], r["kernel"], r["avg_launch_ms"], r["l1_gather"] and round(r["l1_gather"]["frac"],2)))
PY
}
run spp16 --spp 16
run 4k16 --size 3840x2160 --spp 16
run 1m4 --scene 1m --spp 4
run 1m64 --scene 1m --spp 64 --steps 8 --warmup 8
run spp16_b1 --spp 16 --batch 1
run 1m4_b1 --scene 1m --spp 4 --batch 1

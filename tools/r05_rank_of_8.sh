#!/bin/bash
# round 5: wall ms per frame of EVERY rank of an eight-rank tile deal, each emulated alone on this GPU (batches of 8 frames, bench workload):
# the imbalance of the deal as a measured number.  No gather, no RCCL: rank-local work only.
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-r05r8}; mkdir -p $OUT; cd $R
W=${2:-8}
for r in $(seq 0 $((W-1))); do timeout -k 10 120 python3 tools/wall_batch.py $W 8 1920x1080 $r 2>&1 | tail -1 | tee -a $OUT/ranks.log; done
python3 - <<PY | tee -a $OUT/ranks.log
import re
v = [float(re.search(r"([0-9.]+) ms/frame", l).group(1)) for l in open("$OUT/ranks.log") if "ms/frame" in l][-$W:]
print("per_rank_ms", v, "max", max(v), "mean", round(sum(v) / len(v), 4), "imbalance", round(max(v) / (sum(v) / len(v)), 3))
PY

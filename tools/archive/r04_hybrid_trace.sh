#!/bin/bash
# round 4: per-launch kernel durations of one hybrid frame set (which pass costs what) + A/B of the binning experiment on the wavefront pipeline
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-r04ht}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--hybrid --spp 16 --gi-bounces 4 --steps 4 --warmup 2 --cpu-seconds 0 --no-default-camera --no-frame-by-frame --parity-window 0"
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py $ARGS > $OUT/trace.log 2>&1
cp $OUT/trace/*/*kernel_stats.csv $OUT/hybrid_kernel_stats.csv 2>/dev/null; head -8 $OUT/hybrid_kernel_stats.csv | cut -c1-160
python3 - <<PY
import csv, glob, collections
f = glob.glob("$OUT/trace/*/*kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# the last frame: from the last k_hybrid_resolve backwards to the one before
names = [r["Kernel_Name"] for r in rows]
idx = [i for i, n in enumerate(names) if "k_hybrid_resolve" in n]
a, b = idx[-2] + 1, idx[-1] + 1
t0 = int(rows[a]["Start_Timestamp"])
for r in rows[a:b]:
    n = r["Kernel_Name"]
    short = "shade" if "hybrid_shade" in n else "verify" if "hybrid_verify" in n else "note" if "hybrid_note" in n else "trace" if "k_trace" in n else "resolve" if "resolve" in n else n[:30]
    print("%-8s start %8.3f ms  dur %8.3f ms" % (short, (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
PY
rm -rf $OUT/trace

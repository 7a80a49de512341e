#!/bin/bash
# kernel timeline of the as-benchmarked mode (3 lanes, batches of 8): rocprofv3 --kernel-trace, analysed by tools/r03_timeline.py
TAG=${1:-r03t}; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/tl -- python3 $R/tools/wall_batch.py 1 8 > $OUT/tl.log 2>&1; tail -1 $OUT/tl.log
cp $OUT/tl/*/*kernel_trace.csv $OUT/kernel_trace.csv; ls -la $OUT/kernel_trace.csv
python3 $R/tools/r03_timeline.py $OUT/kernel_trace.csv | tee $OUT/timeline.txt

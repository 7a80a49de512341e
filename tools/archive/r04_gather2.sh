#!/bin/bash
# round 4: the quad-cooperative fetch microbenchmark (tools/gather2.hip) -> gpurun_out/<tag>/gather2.txt
TAG=${1:-r04g}; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/gather2_bench tools/gather2.hip 2>/dev/null || exit 1
timeout -k 10 300 /tmp/gather2_bench > $OUT/gather2.txt 2>&1
echo "exit $?"; head -3 $OUT/gather2.txt; wc -l $OUT/gather2.txt

#!/bin/bash
# Run on the MI355X box (through gpurun): collects the rocprofv3 summaries quoted in DESIGN.md / bench.py.
#   tools/collect_profiles.sh <tag>     -> gpurun_out/profiles_<tag>/
TAG=${1:-r01}
OUT=$GRAFT_REPO_ROOT/gpurun_out/profiles_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# 1. kernel trace + stats of the bench command itself
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --cpu-seconds 0 > $OUT/bench_under_rocprof.log 2>&1
cp $OUT/trace/*/*kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null
# 1b. the same with one frame in flight (RT_LANES=1): per-kernel durations without cross-frame overlap
export RT_LANES=1
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace1 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --cpu-seconds 0 --no-default-camera > $OUT/bench_under_rocprof_lanes1.log 2>&1
cp $OUT/trace1/*/*kernel_stats.csv $OUT/kernel_stats_lanes1.csv 2>/dev/null
unset RT_LANES
# 2. PMC passes (counters only), separate passes per counter group, on 4 frames of the same workload
for i in 1 2 3 5 6; do
  case $i in
    1) C="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU";;
    2) C="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_WR";;
    3) C="TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum";;
    5) C="FETCH_SIZE";;
    6) C="WRITE_SIZE";;
  esac
  timeout -k 10 150 rocprofv3 --pmc $C --output-format csv -d $OUT/p$i -- python3 $GRAFT_REPO_ROOT/tools/prof_frames.py --frames 4 > $OUT/p$i.log 2>&1 || echo "pmc set $i failed"
done
python3 $GRAFT_REPO_ROOT/tools/pmc_sum.py $OUT --frames 7 > $OUT/pmc_summary.txt
cat $OUT/pmc_summary.txt | cut -c1-200

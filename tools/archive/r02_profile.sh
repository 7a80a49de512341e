#!/bin/bash
# Round-2 measurement pass, run on the MI355X box through gpurun:  tools/r02_profile.sh <tag> [steps...]
#   steps: tests bench bench1m trace pmc_bunny pmc_1m gather   (default: all)
# Everything lands under gpurun_out/<tag>/; summaries worth keeping are copied into profiles/ by hand (tools/README.md).
TAG=${1:-r02a}; shift
STEPS=${@:-tests bench bench1m trace pmc_bunny pmc_1m gather}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
has() { [[ " $STEPS " == *" $1 "* ]]; }
pmc() {   # pmc <outdir> <counters...> -- <program...>
  local d=$1; shift; local c=(); while [[ "$1" != "--" ]]; do c+=("$1"); shift; done; shift
  mkdir -p $(dirname $d); timeout -k 10 240 rocprofv3 --pmc "${c[@]}" --output-format csv -d $d -- "$@" > $d.log 2>&1 || { echo "pmc pass $d failed"; tail -3 $d.log; }
}
if has tests; then
  (cd $R && timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1); echo "pytest rc=$?"; tail -3 $OUT/pytest.log
fi
if has bench; then
  (cd $R && timeout -k 10 600 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err); echo "bench rc=$?"; cut -c1-600 $OUT/bench.json
fi
if has bench1m; then
  (cd $R && timeout -k 10 600 python3 bench.py --scene 1m --steps 20 --warmup 3 --cpu-seconds 0 --no-default-camera > $OUT/bench_1m.json 2> $OUT/bench_1m.err); echo "bench1m rc=$?"; cut -c1-400 $OUT/bench_1m.json
fi
if has trace; then
  # kernel trace + stats of the bench command itself (frames overlapping on 3 lanes) and with one frame in flight
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 40 --warmup 8 --cpu-seconds 0 --no-default-camera --no-frame-by-frame > $OUT/trace.log 2>&1
  cp $OUT/trace/*/*kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null
  RT_LANES=1 timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace1 -- python3 $R/bench.py --steps 40 --warmup 8 --cpu-seconds 0 --no-default-camera --no-frame-by-frame > $OUT/trace1.log 2>&1
  cp $OUT/trace1/*/*kernel_stats.csv $OUT/kernel_stats_one_frame_in_flight.csv 2>/dev/null
  # frame by frame, one frame in flight: the launches the PMC passes below count (tools/prof_frames.py)
  RT_LANES=1 timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tracepf -- python3 $R/tools/prof_frames.py --frames 4 > $OUT/tracepf.log 2>&1
  cp $OUT/tracepf/*/*kernel_stats.csv $OUT/kernel_stats_frame_by_frame_one_in_flight.csv 2>/dev/null
  RT_LANES=1 timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace1m -- python3 $R/tools/prof_frames.py --scene 1m --frames 4 > $OUT/trace1m.log 2>&1
  cp $OUT/trace1m/*/*kernel_stats.csv $OUT/kernel_stats_1m_one_frame_in_flight.csv 2>/dev/null
  echo "trace done"; ls $OUT/*.csv
fi
SETS=(
 "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU"
 "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_WR"
 "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE"
 "FETCH_SIZE"
 "WRITE_SIZE"
 "TCP_TOTAL_ACCESSES_sum TCP_TAGRAM0_REQ_sum TCP_TAGRAM1_REQ_sum"
 "TA_BUSY_avr"
 "TA_FLAT_READ_WAVEFRONTS_sum"
 "TA_ADDR_STALLED_BY_TC_CYCLES_sum"
 "TA_DATA_STALLED_BY_TC_CYCLES_sum"
 "TCP_TA_TCP_STATE_READ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum"
 "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_REQ_sum"
)
if has pmc_bunny; then
  for i in "${!SETS[@]}"; do RT_LANES=1 pmc $OUT/pmc_bunny/p$i ${SETS[$i]} -- python3 $R/tools/prof_frames.py --frames 4; done
  python3 $R/tools/pmc_sum.py $OUT/pmc_bunny --frames 7 > $OUT/pmc_bunny_summary.txt; cp $OUT/pmc_bunny/traffic.json $OUT/traffic_bunny.json
  cut -c1-300 $OUT/pmc_bunny_summary.txt | grep -A2 "k_trace"
fi
if has pmc_1m; then
  for i in 0 1 2 3 4 11; do RT_LANES=1 pmc $OUT/pmc_1m/p$i ${SETS[$i]} -- python3 $R/tools/prof_frames.py --scene 1m --frames 2; done
  python3 $R/tools/pmc_sum.py $OUT/pmc_1m --frames 5 > $OUT/pmc_1m_summary.txt; cp $OUT/pmc_1m/traffic.json $OUT/traffic_1m.json
  cut -c1-300 $OUT/pmc_1m_summary.txt | grep -A2 "k_trace"
fi
if has gather; then
  $R/tools/gather_bench > $OUT/gather.txt 2>&1; tail -n +1 $OUT/gather.txt | cut -c1-160
  for i in 2 5 6 7 8 10; do pmc $OUT/pmc_gather/p$i ${SETS[$i]} -- $R/tools/gather_bench; done
  python3 $R/tools/gather_pmc.py $OUT/pmc_gather $OUT/gather.txt > $OUT/gather_pmc.txt; cat $OUT/gather_pmc.txt | cut -c1-250
fi

#!/bin/bash
# Round-3 measurement pass, run on the MI355X box through gpurun:  tools/r03_profile.sh <tag> [steps...]
#   steps: trace pmc_bunny pmc_1m   (default: all).  Everything profiles bench.py ITSELF in its timed mode (batches of 8 frames), the
#   program directly after `--`; one counter set per rocprofv3 pass (TCC: FETCH_SIZE and WRITE_SIZE cannot share a pass; one TA counter per pass).
#   Results: gpurun_out/<tag>/ ; tools/r03_summarize.py copies what is kept into profiles/r03_*.
TAG=${1:-r03p}; shift
STEPS=${@:-trace pmc_bunny pmc_1m}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
has() { [[ " $STEPS " == *" $1 "* ]]; }
BENCH="python3 $R/bench.py --steps 40 --warmup 8 --cpu-seconds 0 --no-default-camera --no-frame-by-frame --no-diagnostics --no-run-b"
BENCH1M="python3 $R/bench.py --scene 1m --steps 8 --warmup 8 --cpu-seconds 0 --no-default-camera --no-frame-by-frame --no-diagnostics --no-run-b"
pmc() {   # pmc <outdir> <counters...> -- <program...>
  local d=$1; shift; local c=(); while [[ "$1" != "--" ]]; do c+=("$1"); shift; done; shift
  mkdir -p $(dirname $d); timeout -k 10 400 rocprofv3 --pmc "${c[@]}" --output-format csv -d $d -- "$@" > $d.log 2>&1 || { echo "pmc pass $d failed"; tail -3 $d.log; }
}
if has trace; then
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH > $OUT/trace.log 2>&1
  cp $OUT/trace/*/*kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null
  RT_LANES=1 timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace1 -- $BENCH > $OUT/trace1.log 2>&1
  cp $OUT/trace1/*/*kernel_stats.csv $OUT/kernel_stats_one_launch_set_in_flight.csv 2>/dev/null
  RT_LANES=1 timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace1m -- $BENCH1M > $OUT/trace1m.log 2>&1
  cp $OUT/trace1m/*/*kernel_stats.csv $OUT/kernel_stats_1m_one_launch_set_in_flight.csv 2>/dev/null
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
 "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_REQ_sum"
)
PICK=${R03_SETS:-0 1 2 3 4 5 6 7 8}
if has pmc_bunny; then
  for i in $PICK; do RT_LANES=1 pmc $OUT/pmc_bunny/p$i ${SETS[$i]} -- $BENCH; done
  python3 $R/tools/r03_pmc_sum.py $OUT/pmc_bunny > $OUT/pmc_bunny_summary.txt; cut -c1-400 $OUT/pmc_bunny_summary.txt | grep -A3 "k_trace"
fi
if has pmc_1m; then
  for i in 0 1 2 3 4; do RT_LANES=1 pmc $OUT/pmc_1m/p$i ${SETS[$i]} -- $BENCH1M; done
  python3 $R/tools/r03_pmc_sum.py $OUT/pmc_1m > $OUT/pmc_1m_summary.txt; cut -c1-400 $OUT/pmc_1m_summary.txt | grep -A3 "k_trace"
fi

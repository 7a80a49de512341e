#!/bin/bash
# where the memory pipeline of the traversal launches waits: TA / TCP / TD busy and stall counters, L1-miss latency, address translation.
# One small counter set per rocprofv3 pass over bench.py in its timed mode (batches of 8, RT_LANES=1); each pass carries GRBM_GUI_ACTIVE as the clock.
TAG=${1:-r03stall}; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $R/bench.py --steps 40 --warmup 8 --cpu-seconds 0 --no-default-camera --no-frame-by-frame --no-diagnostics"
BENCH1M="python3 $R/bench.py --scene 1m --steps 8 --warmup 8 --cpu-seconds 0 --no-default-camera --no-frame-by-frame --no-diagnostics"
SETS=(
 "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE"
 "TCP_TCP_LATENCY_sum TCP_TOTAL_ACCESSES_sum GRBM_GUI_ACTIVE"
 "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum GRBM_GUI_ACTIVE"
 "TA_TA_BUSY_sum GRBM_GUI_ACTIVE"
 "TA_ADDR_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE"
 "TA_DATA_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE"
 "TD_TD_BUSY_sum TD_TC_STALL_sum GRBM_GUI_ACTIVE"
 "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum GRBM_GUI_ACTIVE"
 "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum GRBM_GUI_ACTIVE"
 "TCP_GATE_EN1_sum TCP_GATE_EN2_sum GRBM_GUI_ACTIVE"
 "TA_ADDR_STALLED_BY_TD_CYCLES_sum GRBM_GUI_ACTIVE"
)
run() {   # run <dir> <bench...>
  local d=$1; shift
  for i in ${!SETS[@]}; do
    mkdir -p $d
    timeout -k 10 300 rocprofv3 --pmc ${SETS[$i]} --output-format csv -d $d/p$i -- "$@" > $d/p$i.log 2>&1 || { echo "pass $i failed"; tail -2 $d/p$i.log; }
  done
  python3 $R/tools/r03_pmc_sum.py $d > $d/summary.txt 2>/dev/null; grep -A1 "^k_trace" $d/summary.txt | cut -c1-1500
}
RT_LANES=1 run $OUT/bunny $BENCH
RT_LANES=1 run $OUT/1m $BENCH1M

#!/bin/bash
# usage: tools/pmc.sh <outdir-under-gpurun_out> <set-ids e.g. "1 2"> -- <args for tools/prof_frames.py>
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; SETS="$2"; shift; shift; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
declare -A S
S[1]="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU"
S[2]="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_WR"
S[3]="TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum"
S[4]="TA_BUSY_avr"                     # one TA-block counter per pass: two or more abort rocprofv3 on gfx950 ("exceeds the capabilities of the hardware")
S[10]="TA_FLAT_READ_WAVEFRONTS_sum"
S[11]="TA_ADDR_STALLED_BY_TC_CYCLES_sum"
S[12]="TA_DATA_STALLED_BY_TC_CYCLES_sum"
S[13]="TCP_TA_TCP_STATE_READ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum"
S[14]="GRBM_GUI_ACTIVE"
S[5]="FETCH_SIZE"
S[6]="WRITE_SIZE"
S[7]="TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum"
S[8]="TA_TA_BUSY_sum"
S[9]="TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum"
for i in $SETS; do
  timeout -k 10 150 rocprofv3 --pmc ${S[$i]} --output-format csv -d $OUT/p$i -- python3 $GRAFT_REPO_ROOT/tools/prof_frames.py "$@" > $OUT/p$i.log 2>&1 || { echo "set $i failed"; tail -3 $OUT/p$i.log; }
done
python3 $GRAFT_REPO_ROOT/tools/pmc_sum.py $OUT

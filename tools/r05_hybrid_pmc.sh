#!/bin/bash
# round 5: counters of the hybrid extension's shading kernel on run B as written (1080p, 16 spp, 4 bounces).
# One small counter set per rocprofv3 pass over bench.py --hybrid; the program itself after "--"; no tracing options beside --pmc.
TAG=${1:-r05hp}; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--hybrid --spp 16 --gi-bounces 4 --steps 3 --warmup 1 --cpu-seconds 0 --no-default-camera --no-frame-by-frame --parity-window 0 --no-diagnostics"
SETS=(
 "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE"
 "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"
 "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_FLAT GRBM_GUI_ACTIVE"
 "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_FLAT GRBM_GUI_ACTIVE"
 "FETCH_SIZE GRBM_GUI_ACTIVE"
 "WRITE_SIZE GRBM_GUI_ACTIVE"
 "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum GRBM_GUI_ACTIVE"
 "SQ_INSTS_VALU_TRANS SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 GRBM_GUI_ACTIVE"
)
for i in ${!SETS[@]}; do
  timeout -k 10 200 rocprofv3 --pmc ${SETS[$i]} --output-format csv -d $OUT/p$i -- python3 $R/bench.py $ARGS > $OUT/p$i.log 2>&1 || { echo "pass $i failed"; tail -2 $OUT/p$i.log; }
done
python3 $R/tools/pmc_sum.py $OUT 2>/dev/null | grep -A2 "hybrid_shade\|hybrid_verify\|k_trace" | cut -c1-2500 | tee $OUT/summary.txt
rm -rf $OUT/p*/

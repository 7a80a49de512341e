#!/bin/bash
# other BASELINE configurations + emulated tile-parallel ranks (one GPU).  gpurun_out/<tag>/
TAG=${1:-r03c}; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
for w in 1 2 4 8; do python3 tools/wall_batch.py $w 8 | tail -1 | tee -a $OUT/ranks.log; done
for w in 1 8; do python3 tools/wall.py $w | tail -1 | tee -a $OUT/ranks.log; done
bash tools/other_configs.sh $TAG 2>&1 | tee $OUT/configs.log

#!/bin/bash
# Quantised any-hit nodes under test: the whole GPU suite with RT_QNODES=2 (+ the SAH tree, + small chunks), a long fuzz run, then bench lines of the 1 M scene.
TAG=${1:-r04qn}; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
for V in "RT_QNODES=2" "RT_QNODES=1 RT_ANYHIT_TREE=sah RT_CHUNK=64" "RT_QNODES=2 RT_QUEUE_BUDGET_MB=4 RT_LANES=2"; do
  echo "=== [$V]" | tee -a $OUT/qn.log
  env $V timeout -k 10 900 python3 -m pytest tests -x -q -m gpu --deselect tests/test_bench_contract.py 2>&1 | tail -1 | tee -a $OUT/qn.log
done
RT_QNODES=2 RT_FUZZ_CASES=1500 timeout -k 10 900 python3 -m pytest tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -1 | tee -a $OUT/qn.log
for V in RT_QNODES=0 RT_QNODES=-1 RT_QNODES=0 RT_QNODES=-1; do
  env $V python3 bench.py --scene 1m --steps 8 --warmup 8 --cpu-seconds 0 --no-default-camera --no-frame-by-frame --no-run-b > $OUT/b1m_$V.json 2> $OUT/b1m_$V.err
  python3 -c "import json,sys; d=json.load(open('$OUT/b1m_$V.json')); print('$V', d['ms_per_step'], d['parity']['rmse'], d['parity'].get('bit_diff'))" | tee -a $OUT/qn.log
done

#!/bin/bash
# round 4: bench.py contract tests (parity block, multi_gpu block) + one default bench line
TAG=${1:-r04b}; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
timeout -k 10 900 python3 -m pytest tests/test_bench_contract.py tests/test_c_abi.py -x -q > $OUT/pytest.log 2>&1; echo "pytest rc $?"; tail -5 $OUT/pytest.log
timeout -k 10 400 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc $?"; python3 - <<PY
import json
d=json.load(open("$OUT/bench.json"))
print(d["ms_per_step"], d["value"], d["parity"], d["roofline"]["frac"], d["roofline"]["l1_gather"]["frac"])
PY

#!/bin/bash
# Round-3 first contact: GPU test suite, bench line, traversal statistics of both scenes.  gpurun_out/<tag>/
TAG=${1:-r03a}; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
(timeout -k 10 1500 python3 -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1); echo "pytest rc=$?"; tail -5 $OUT/pytest.log
(timeout -k 10 600 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err); echo "bench rc=$?"; cut -c1-1500 $OUT/bench.json; tail -3 $OUT/bench.err
RT_LANES=1 RT_TRACE_STATS=1 timeout -k 10 300 python3 tools/prof_frames.py --frames 16 --batch 8 > $OUT/stats_bunny.log 2>&1; tail -6 $OUT/stats_bunny.log | cut -c1-700
RT_LANES=1 RT_TRACE_STATS=1 timeout -k 10 300 python3 tools/prof_frames.py --frames 3 --scene 1m > $OUT/stats_1m.log 2>&1; tail -6 $OUT/stats_1m.log | cut -c1-700

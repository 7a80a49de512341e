#!/bin/bash
# end-of-round sequence on the GPU box: full GPU suite, bench line, then the profile passes that stamp profiles/r03_traffic_*.json
TAG=${1:-r03z}; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
(timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1); echo "pytest rc=$?"; tail -2 $OUT/pytest.log
R03_SETS="2 3 4" bash tools/r03_profile.sh $TAG pmc_bunny pmc_1m > $OUT/profile.log 2>&1; tail -3 $OUT/profile.log | cut -c1-200

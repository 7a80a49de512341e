#!/bin/bash
# round 4: register budget of the hybrid shading kernel (RT_HYBRID_WAVES = waves per SIMD of k_hybrid_shade): ms per frame of run B, 1080p / 16 spp / 4 bounces
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-r04hw}; mkdir -p $OUT; cd $R
for W in 5 4 6 3 5 4; do
  RT_HYBRID_WAVES=$W timeout -k 10 300 python3 bench.py --hybrid --spp 16 --gi-bounces 4 --steps 6 --warmup 2 --cpu-seconds 0 --no-default-camera --no-frame-by-frame --no-diagnostics --no-run-b --parity-window 32x16 > $OUT/w$W.json 2> $OUT/w$W.err
  python3 -c "import json; d=json.loads(open('$OUT/w$W.json').read().strip().splitlines()[-1]); print('waves $W', round(d['ms_per_step'],2), d['parity']['ok'], d['parity'].get('bit_diff'))" | tee -a $OUT/hw.log
done

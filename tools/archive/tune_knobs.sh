#!/bin/bash
# Knob sweep on the bench frame (run through gpurun): wall ms/frame with the default frame lanes.  RT_CHUNK 0 = automatic.
for cfg in "32 16 0 2" "24 16 0 2" "40 16 0 2" "32 8 0 2" "32 24 0 2" "32 16 128 2" "32 16 256 2" "32 16 512 2" "32 16 0 4" "24 12 0 2" "28 16 0 2" "36 16 0 2"; do set -- $cfg
  echo -n "refill=$1 minSearch=$2 run=$3 leafb=$4: "
  RT_REFILL_MIN=$1 RT_MIN_SEARCH=$2 RT_CHUNK=$3 RT_LEAFB=$4 timeout -k 5 120 python3 tools/wall.py ${W:-1} 2>/dev/null | tail -1
done

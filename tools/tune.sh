#!/bin/bash
for cfg in "64 2 0 32" "64 2 1 32" "64 4 1 32" "32 4 1 16" "16 4 1 8" "16 2 1 8" "32 2 1 16"; do set -- $cfg
  echo -n "run=$1 leafb=$2 overlap=$3 refill=$4: "
  RT_CHUNK=$1 RT_LEAFB=$2 RT_OVERLAP=$3 RT_REFILL_MIN=$4 timeout -k 5 120 python3 tools/prof_frames.py --frames 8 --world ${W:-8} | tail -1 | cut -c1-225
done

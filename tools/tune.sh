#!/bin/bash
for lb in 1 2 4; do
  echo -n "leafb=$lb: "
  RT_LEAFB=$lb timeout -k 5 120 python3 tools/prof_frames.py --frames 8 "$@" | tail -1 | cut -c1-215
done

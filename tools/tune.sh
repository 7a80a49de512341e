#!/bin/bash
for ch in 64 128 256; do for rm in 24 32 48; do
  echo -n "run=$ch refill=$rm: "
  RT_CHUNK=$ch RT_REFILL_MIN=$rm timeout -k 5 120 python3 tools/prof_frames.py --frames 6 "$@" | tail -1 | cut -c1-215
done; done

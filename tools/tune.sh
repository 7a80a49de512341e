#!/bin/bash
for ch in 512 2048; do for rm in 28 36; do for ms in 8 16; do
  echo -n "chunk=$ch refill=$rm ms=$ms: "
  RT_CHUNK=$ch RT_REFILL_MIN=$rm RT_MIN_SEARCH=$ms timeout -k 5 120 python3 tools/prof_frames.py --frames 4 "$@" | tail -1
done; done; done

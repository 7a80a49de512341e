#!/bin/bash
for sz in 480x270 960x540 1920x1080 3840x2160; do
  echo -n "$sz: "; python3 tools/prof_frames.py --frames 4 --size $sz "$@" | tail -1
done

#!/bin/bash
for c in 0 64 128 192 256; do echo -n "primary run=$c: "; RT_CHUNK_PRIMARY=$c python3 tools/wall.py 1 | tail -1; RT_CHUNK_PRIMARY=$c RT_LANES=1 python3 tools/prof_frames.py --frames 6 2>&1 | tail -1 | cut -c1-60; done

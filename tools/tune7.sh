#!/bin/bash
for spp in 4 16; do for mb in 8192 2048 1024 512 256 128 64; do echo -n "spp=$spp budget=${mb}MB: "; RT_QUEUE_BUDGET_MB=$mb timeout -k 5 120 python3 tools/prof_frames.py --frames 6 --spp $spp 2>/dev/null | tail -1 | cut -c1-230; done; done

#!/bin/bash
# frame lanes vs hardware queues (ROCm maps HIP streams onto GPU_MAX_HW_QUEUES hardware queues, default 4)
for q in 4 8 16; do for l in 4 6 8; do echo -n "world=${W:-8} hwq=$q lanes=$l: "; GPU_MAX_HW_QUEUES=$q RT_LANES=$l python3 tools/wall.py ${W:-8} 2>/dev/null | tail -1; done; done

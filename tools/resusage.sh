#!/bin/bash
# Compact per-kernel resource usage (VGPRs / occupancy / LDS) of rt_wave.hip; optional grep filter as $1.
cd /tmp && /opt/rocm/bin/hipcc -O3 -std=c++17 -ffp-contract=off -fno-fast-math --offload-arch=gfx950 -c /root/repo/opengl-raytracing_amd/csrc/rt_wave.hip -o /tmp/w.o -Rpass-analysis=kernel-resource-usage 2>&1 \
 | grep -E "Function Name|  VGPRs:|AGPRs|Occupancy|LDS Size|ScratchSize" | sed 's/.*remark: [^ ]* *//; s/ \[-Rpass.*//' | paste - - - - - - \
 | sed 's/Function Name: _ZN12_GLOBAL__N_1//' | grep -E "${1:-.}"

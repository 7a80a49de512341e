// rt_wave.hpp -- interface between rt_api.hip and the wavefront pipeline (rt_wave.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "rt_frame.hpp"

struct RtContext;
struct RtWave;

RtWave *rt_wave_create(int computeUnits);
void rt_wave_destroy(RtWave *w);
const char *rt_wave_error(const RtWave *w);
// Renders one frame of a BVH scene into `tg`.  `host` is the host copy of *dFrame.
int rt_wave_render(RtWave *w, RtContext *ctx, hipStream_t stream, const rtd::DevFrame *dFrame, const rtd::DevFrame &host,
                   rtd::Targets tg, unsigned long long *counters, bool count, int treeDepth);

// stage timing hooks (rt_api.hip); stage ids index rt_stage_name()
void rt_stage_begin(RtContext *c, int stage);
void rt_stage_end(RtContext *c, int stage, int launches);

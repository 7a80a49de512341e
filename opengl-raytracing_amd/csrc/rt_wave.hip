// rt_wave.hip -- wavefront pipeline (placeholder while the megakernel path is brought up).
#include "rt_wave.hpp"

#include <string>

#include "../../include/rt_mi355.h"

struct RtWave { std::string err; int cus; };
RtWave *rt_wave_create(int cus) { RtWave *w = new RtWave(); w->cus = cus; return w; }
void rt_wave_destroy(RtWave *w) { delete w; }
const char *rt_wave_error(const RtWave *w) { return w->err.c_str(); }
int rt_wave_render(RtWave *w, RtContext *, hipStream_t, const rtd::DevFrame *, const rtd::DevFrame &, rtd::Targets, unsigned long long *, bool, int) {
    w->err = "wavefront pipeline not built";
    return RT_ERR_UNSUPPORTED;
}

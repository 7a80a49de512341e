// rt_wave.hpp -- interface between rt_api.hip and the wavefront pipeline (rt_wave.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "rt_frame.hpp"

struct RtContext;
struct RtWave;
struct RtHybrid;

// Ray-queue arenas of a context, shared by its frame lanes (round 4): lane l uses arena l % n and waits for the arena's previous user (another
// lane's batch, from its first shading launch to its last) through an event.  n == number of lanes: every lane its own arena, as in rounds 1-3.
struct RtArenaPool;
RtArenaPool *rt_arena_pool_create(int arenas);
void rt_arena_pool_destroy(RtArenaPool *p);
size_t rt_arena_pool_bytes(const RtArenaPool *p);
int rt_arena_pool_count(const RtArenaPool *p);
size_t rt_wave_frame_bytes(const RtWave *w);   // per-lane frame arrays (candidates, hits, pre-resolve stash)
size_t rt_hybrid_arena_bytes(const RtHybrid *h);

RtWave *rt_wave_create(int computeUnits, RtArenaPool *pool = nullptr, int lane = 0);
void rt_wave_destroy(RtWave *w);
const char *rt_wave_error(const RtWave *w);
// Renders one frame of a BVH scene into `tg` on `stream`.  `host` is the host copy of *dFrame.  Only the final temporal
// resolve waits for `evPrevDone` (the previous frame's completion event, may be null).  cacheResident: the BVH arrays fit the 32 MB of L2
// (sizes the persistent grids, see rt_wave_render).
int rt_wave_render(RtWave *w, RtContext *ctx, hipStream_t stream, const rtd::DevFrame *dFrame, const rtd::DevFrame &host,
                   rtd::Targets tg, unsigned long long *counters, bool count, int treeDepth, hipEvent_t evPrevDone, bool cacheResident);

// tallies accumulated since the last reset, 16 words: [0] candidate pixels [1] hit pixels [2] primary [3] shadow+AO
// [4] bounce [5] bounce-shadow rays actually traversed, [6] frames, [8..10] 16-byte gather loads issued by the primary /
// any-hit / bounce traversal launches, [11..13] the same after merging the lanes of a wave that read the same record (only
// counted by the diagnostic kernels, RT_TRACE_STATS=1; 0 otherwise)
int rt_wave_traced(RtWave *w, hipStream_t stream, unsigned long long *out16, bool reset);

// Closest-hit traversal of the rays listed in idx[0 .. *count) (queue addresses into o / d) with the persistent kernel of the wavefront
// pipeline; results to outT / outTri at the same address.  heads: rt_wave_head_words() zeroed uint32 cursor words.
void rt_wave_trace_closest_indexed(hipStream_t st, int cus, int treeDepth, const rtd::DevFrame *dFrame, const rtd::DevScene &hostScene, const uint32_t *idx,
                                   const uint32_t *count, const float4 *o, const float4 *d, float *outT, int *outTri, uint32_t *heads);
// The same over a dense array of records o[r] / d[r], r < min(*count, cap); the answer of record r goes to outT / outTri at dst[r].
// Nothing is traced when *flags has bit 2 or 4 set (rt_hybrid.hip: a pass that outgrew its arrays left the queue incomplete).
void rt_wave_trace_closest_compact(hipStream_t st, int cus, int treeDepth, const rtd::DevFrame *dFrame, const rtd::DevScene &hostScene, const float4 *o, const float4 *d,
                                   const uint32_t *dst, const uint32_t *count, const uint32_t *flags, uint32_t cap, float *outT, int *outTri, uint32_t *heads,
                                   uint32_t capOut = 0);   // capOut != 0 (RT_HYBRID_CHECK): entries of outT / outTri, checked before every store
void rt_wave_debug_trace(hipStream_t st, int cus, int treeDepth, const rtd::DevFrame *dFrame, const rtd::DevScene &hostScene, bool any, const float4 *o, const float4 *d,
                         const float *tm, const uint32_t *liveCount, uint32_t n, float *outT, int *outTri, uint8_t *outOcc, uint32_t *heads);
size_t rt_wave_head_words();

// rt_hybrid.hip -- EXTENSION: the hybrid scene (analytic objects + mesh, N diffuse bounces) in stages: shading passes that replay answered mesh
// queries and queue the open ones, persistent traversal launches in between.  Bit-identical to the megakernel's hybrid frames.
struct RtHybrid;
RtHybrid *rt_hybrid_create(int computeUnits);
void rt_hybrid_destroy(RtHybrid *h);
const char *rt_hybrid_error(const RtHybrid *h);
int rt_hybrid_render(RtHybrid *h, RtContext *ctx, hipStream_t stream, const rtd::DevFrame *dFrame, const rtd::DevFrame &host, rtd::Targets tg, int treeDepth,
                     hipEvent_t evPrevDone);

// stage timing hooks (rt_api.hip); stage ids index rt_stage_name()
void rt_stage_begin(RtContext *c, int stage, hipStream_t on = nullptr);   // on == nullptr: the context's stream
void rt_stage_end(RtContext *c, int stage, int launches, hipStream_t on = nullptr);

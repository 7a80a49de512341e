// rt_api.hip -- C ABI of librt_mi355.so (include/rt_mi355.h): context, uploads, frame loop, readback.
//
// The context owns one HIP stream and all device memory.  There is no CPU rendering path: without
// a usable HIP device rt_create fails with RT_ERR_NO_DEVICE and nothing else can be called.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include <dlfcn.h>
#include <rccl/rccl.h>   // types and prototypes only: RCCL is bound at run time (rccl_api below), single-GPU users never load it

#include "../../include/rt_mi355.h"
#include "rt_frame.hpp"
#include "rt_wave.hpp"

using namespace rtd;

static thread_local std::string g_createError;

#define RT_MAX_LANES 8
constexpr size_t kQNodesAbove = (size_t)4 << 20;   // bytes of 112-byte any-hit nodes beyond which the quantised nodes are built and walked
constexpr int kDefaultArenas = 2;   // ray-queue arenas shared by the frame lanes (rt_wave.hpp RtArenaPool; measured in profiles/r04_experiments.txt)
struct StageEvent { int stage; hipEvent_t a, b; };

struct RtContext {
    RtDeviceConfig cfg{};
    // Frames in flight: frame f runs on lane f % nLanes (own stream, frame descriptor, ray-queue arenas, COLOR0 buffer), so
    // up to nLanes consecutive frames overlap everywhere except at the temporal resolve.  stream == lanes[0]: every
    // non-frame operation runs there after a sync of all lanes.
    int nLanes = 3;
    bool serialFrames = false;          // RT_LANES=1: a frame starts when its predecessor has finished
    hipStream_t lanes[RT_MAX_LANES] = {};
    hipStream_t stream = nullptr;
    hipStream_t lastStream = nullptr;    // stream of the most recent frame (gather / assemble are ordered behind it)
    hipEvent_t evDone[RT_MAX_LANES] = {};   // the frame on lane i has written its targets
    std::string err;
    // scene
    float4 *dWNodes = nullptr, *dW4 = nullptr, *dTris = nullptr, *dWNodesW = nullptr, *dPairs = nullptr;
    float4 *dQ4 = nullptr, *dLeafBox = nullptr;   // RT_QNODES: quantised any-hit nodes + the leaves' exact boxes
    float4 *dWF = nullptr;           // fused closest-hit records (round 5), null when the tree's boxes are not the unions of their children's
    float4 *dIN4 = nullptr, *dIQ4 = nullptr, *dILeafBox = nullptr;   // ... and the any-hit walk's four-wide records, exact (96 B) and quantised (48 B + the leaves' exact boxes by ordinal)
    float4 *dIN2 = nullptr, *dIPairs = nullptr;   // implicit records (round 5): 48-byte two-child records without references + the pair records in leaf order; null unless every leaf sits at depth implD
    int implD = 0, implR = 0;
    size_t nFused = 0;
    int sceneFlags = 0;              // RT_SCENE_* bits of RtSceneInfo.flags
    int rootRefW = 0;
    void *dHistAll[RT_MAX_LANES] = {};      // tile-parallel + moving camera: every rank's COLOR0 block of the frame a lane rendered
    bool histExchanged[RT_MAX_LANES] = {};
    uchar4 *dEnv = nullptr;
    int envSize = 0;
    int nNodes = 0, nTris = 0, nInner = 0, rootRef = 0, rootRef4 = 0, treeDepth = 0;
    size_t nWide4 = 0, nPairs = 0;   // records in dW4 / dPairs
    size_t nLeafBoxes = 0;           // leaves with an exact box in dLeafBox (quantised any-hit nodes)
    uint32_t leafBoxMagic = 0;       // dLeafBox index of a leaf = (first pair record * magic) >> 32 (0: = first)
    int anyStack = 0;                // stack entries of the any-hit walk (0: from the binary depth)
    float rootMin[3] = {0, 0, 0}, rootMax[3] = {0, 0, 0};
    // frame state
    FrameGeom g{};
    bool sized = false;
    uint2 *dColor[RT_MAX_LANES] = {};   // COLOR0 ring: frame f writes [f % nLanes], reads [(f-1) % nLanes]
    // motion / position / normal are ringed like COLOR0: a gather (or any other reader) of frame f's targets runs on lane f's
    // stream and must not see frame f+1's stores, which run on another stream
    uint32_t *dMotion[RT_MAX_LANES] = {};
    uint2 *dGPos[RT_MAX_LANES] = {}, *dGNrm[RT_MAX_LANES] = {};
    size_t nSlots = 0;
    int frameIndex = 0, writeIdx = 0;     // include/render/accum.h:125-138
    bool haveFrameState = false;
    float prevVP[16];
    DevFrame *dFrame[RT_MAX_LANES] = {};
    unsigned long long *dCounters = nullptr;
    void *dStaging = nullptr;
    size_t stagingBytes = 0;
    RtWave *wave[RT_MAX_LANES] = {};
    RtArenaPool *arenaPool = nullptr;   // ray-queue arenas shared by the lanes' wavefront pipelines
    RtHybrid *hybrid[RT_MAX_LANES] = {};   // EXTENSION: staged hybrid pipeline, created on first use
    int cus = 256;
    int giBounces = 1;   // EXTENSION, rt_set_extension
    int envFilter = 0;   // rt_set_extension: cube-map filter model (0 exact fp32 weights, 1 coordinates rounded to 1/256 texel)
    // tile-parallel exchange owned by the library (rt_comm.cpp): RCCL communicator + per-lane gather buffers on the gathering rank
    void *comm = nullptr;                               // ncclComm_t
    void *dGathered[RT_MAX_LANES][4] = {};              // [lane][target]: worldSize blocks, rank-major
    void *dAssembled[RT_MAX_LANES][4] = {};             // [lane][target]: row-major frame of halfs
    int gatheredLane[4] = {-1, -1, -1, -1};             // lane whose frame rt_gather_frame(which) gathered last
    // timing
    bool timing = false;
    std::vector<StageEvent> pending;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> freeEvents;
    uint64_t gathers = 0, gatherBytes = 0, historyExchanges = 0;   // rt_comm_info
    double stageMs[RT_MAX_STAGES] = {0};
    uint64_t stageLaunches[RT_MAX_STAGES] = {0};
    int timedFrames = 0;
};

static int fail(RtContext *c, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) c->err = buf; else g_createError = buf;
    return code;
}
// No C++ exception crosses the C ABI: std::bad_alloc etc. from the host-side repacking become status codes.
template <class F> static int guarded(RtContext *c, const char *what, F &&body) {
    try { return body(); }
    catch (const std::bad_alloc &) { return fail(c, RT_ERR_IO, "%s: out of host memory", what); }
    catch (...) { return fail(c, RT_ERR_INVALID, "%s: unexpected exception", what); }
}
static hipError_t sync_all(RtContext *c) {
    hipError_t e = hipSuccess;
    for (int i = 0; i < c->nLanes; ++i)
        if (c->lanes[i]) { hipError_t ei = hipStreamSynchronize(c->lanes[i]); if (e == hipSuccess) e = ei; }
    return e;
}
#define HIP_TRY(c, expr)                                                                                  \
    do {                                                                                                  \
        hipError_t e_ = (expr);                                                                           \
        if (e_ != hipSuccess) return fail((c), RT_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

static const char *kStageNames[RT_MAX_STAGES] = {"mega",     "primary", "trace_primary",   "post_primary", "gen_direct", "trace_shadow",
                                                 "trace_gi", "gen_gi",  "resolve",         "combine",      "assemble",   "present",
                                                 "gather", "trace_ao"};   // gather: the whole of rt_gather_frame on its stream (copy / send / recv / un-tiling; "assemble" lies inside it)

// ------------------------------------------------------------------------------------------------
namespace {

__global__ void k_untile(const void *src, void *dst, FrameGeom g, int channels, int toF32) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= g.W * g.H) return;
    int x = i % g.W, y = i / g.W;
    int s = slot_of_pixel(g, x, y);
    const uint16_t *in = (const uint16_t *)src + (size_t)(s < 0 ? 0 : s) * channels;
    for (int c = 0; c < channels; ++c) {
        uint16_t h = (s < 0) ? (uint16_t)0 : in[c];
        if (toF32) ((float *)dst)[(size_t)i * channels + c] = f16_bits_to_f32(h);
        else ((uint16_t *)dst)[(size_t)i * channels + c] = h;
    }
}

// Row-major frame of halfs -> this rank's tile-major slots (inverse of k_untile).
__global__ void k_tile(const void *src, void *dst, FrameGeom g, int channels) {
    int slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= g.nLocalTiles * 256) return;
    int x, y;
    if (!pixel_of_slot(g, slot >> 8, slot & 255, x, y)) return;
    const uint16_t *in = (const uint16_t *)src + ((size_t)y * g.W + x) * channels;
    uint16_t *out = (uint16_t *)dst + (size_t)slot * channels;
    for (int c = 0; c < channels; ++c) out[c] = in[c];
}

// Gathered blocks (rank-major, blockBytes each, tile-major inside) -> row-major frame of halfs.
__global__ void k_assemble(const void *gathered, void *dst, FrameGeom g, int channels, size_t blockBytes) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= g.W * g.H) return;
    int x = i % g.W, y = i / g.W;
    int tx = x >> 4, ty = y >> 4;
    int t = tile_index(g, tx, ty);
    int owner = t % g.world, local = t / g.world;
    int lx = x & 15, ly = y & 15;
    int q = (lx >> 3) | ((ly >> 3) << 1);
    size_t slot = (size_t)local * 256 + q * 64 + (ly & 7) * 8 + (lx & 7);
    const uint16_t *in = (const uint16_t *)((const char *)gathered + (size_t)owner * blockBytes) + slot * channels;
    uint16_t *out = (uint16_t *)dst + (size_t)i * channels;
    for (int c = 0; c < channels; ++c) out[c] = in[c];
}

__global__ void k_debug_eval(int op, const float *a, const float *b, const float *c, uint32_t *out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float x = a[i], y = b ? b[i] : 0.0f, z = c ? c[i] : 0.0f;
    float s, co;
    uint32_t r = 0;
    switch (op) {
        case 0: sincosr(x, s, co); r = f2u(s); break;
        case 1: sincosr(x, s, co); r = f2u(co); break;
        case 2: r = f2u(exp2r(x)); break;
        case 3: r = f2u(log2r(x)); break;
        case 4: r = f2u(powr(x, y)); break;
        case 5: r = f32_to_f16_bits(x); break;
        case 6: r = rand_bits(x, y, (int)z); break;
        case 7: r = f2u(x / y); break;
        case 8: r = f2u(__builtin_sqrtf(x)); break;
        case 9: r = f2u(1.0f / __builtin_sqrtf(x)); break;
        default: break;
    }
    out[i] = r;
}

__global__ __launch_bounds__(256) void k_debug_trace(DevScene sc, int kind, const float *o, const float *d, const float *tMax, float eps,
                                                     float inf, float *out7, int n) {
    __shared__ StackEntry stack[4 * 32 * 64];
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    StackEntry *stk = &stack[(threadIdx.x >> 6) * 32 * 64 + (threadIdx.x & 63)];
    if (i >= n) return;
    Work w;
    work_zero(w);
    V3 ro = ld3(o + (size_t)i * 3), rd = ld3(d + (size_t)i * 3);
    float *out = out7 + (size_t)i * 7;
    if (kind == 0) {
        float t;
        int tri;
        bool hit = bvh_closest<false>(sc, ro, rd, eps, inf, stk, t, tri, w);
        out[0] = hit ? t : inf;
        if (hit) {
            V3 p = ro + rd * t, nn = tri_normal(sc, tri);
            out[1] = p.x; out[2] = p.y; out[3] = p.z; out[4] = nn.x; out[5] = nn.y; out[6] = nn.z;
        } else {
            for (int k = 1; k < 7; ++k) out[k] = 0.0f;
        }
    } else {
        out[0] = bvh_anyhit<false>(sc, ro, rd, eps, tMax[i], stk, w) ? 1.0f : 0.0f;
        for (int k = 1; k < 7; ++k) out[k] = 0.0f;
    }
}

DevScene make_dev_scene(const RtContext *c) {
    DevScene s;
    s.wnodes = c->dWNodes;
    s.w4 = c->dW4;
    s.q4 = c->dQ4;
    s.leafBox = c->dLeafBox;
    s.leafBoxMagic = c->leafBoxMagic;
    s.wnodesW = c->dWNodesW;
    s.wF = c->dWF;
    s.iN2 = c->dIN2; s.iPairs = c->dIPairs; s.implD = c->implD; s.implR = c->implR;
    s.iN4 = c->dIN4; s.iQ4 = c->dIQ4; s.iLeafBox = c->dILeafBox;
    s.pairs = c->dPairs;
    s.rootRefW = c->rootRefW;
    s.tris = c->dTris;
    s.env = c->dEnv;
    s.envSize = c->envSize;
    s.envFilter = c->envFilter;
    s.rootRef = c->rootRef;
    s.rootRef4 = c->rootRef4;
    s.hasBVH = (c->nNodes > 0 && c->nTris > 0) ? 1 : 0;
    s.anyStack = c->anyStack;
    std::memcpy(s.rootMin, c->rootMin, 12);
    std::memcpy(s.rootMax, c->rootMax, 12);
    return s;
}

void free_gather_buffers(RtContext *c) {
    for (int l = 0; l < RT_MAX_LANES; ++l)
        for (int w = 0; w < 4; ++w) {
            if (c->dGathered[l][w]) (void)hipFree(c->dGathered[l][w]);
            if (c->dAssembled[l][w]) (void)hipFree(c->dAssembled[l][w]);
            c->dGathered[l][w] = c->dAssembled[l][w] = nullptr;
        }
    for (int w = 0; w < 4; ++w) c->gatheredLane[w] = -1;
}

void free_targets(RtContext *c) {
    for (int i = 0; i < RT_MAX_LANES; ++i) { if (c->dColor[i]) (void)hipFree(c->dColor[i]); c->dColor[i] = nullptr; }
    for (int i = 0; i < RT_MAX_LANES; ++i) { if (c->dHistAll[i]) (void)hipFree(c->dHistAll[i]); c->dHistAll[i] = nullptr; c->histExchanged[i] = false; }
    for (int i = 0; i < RT_MAX_LANES; ++i) {
        if (c->dMotion[i]) (void)hipFree(c->dMotion[i]);
        if (c->dGPos[i]) (void)hipFree(c->dGPos[i]);
        if (c->dGNrm[i]) (void)hipFree(c->dGNrm[i]);
        c->dMotion[i] = nullptr; c->dGPos[i] = c->dGNrm[i] = nullptr;
    }
    free_gather_buffers(c);
    c->sized = false;
}

int ensure_staging(RtContext *c, size_t bytes) {
    if (c->stagingBytes >= bytes) return RT_OK;
    if (c->dStaging) (void)hipFree(c->dStaging);
    c->dStaging = nullptr; c->stagingBytes = 0;
    HIP_TRY(c, hipMalloc(&c->dStaging, bytes));
    c->stagingBytes = bytes;
    return RT_OK;
}

int last_lane(const RtContext *c) { return (c->writeIdx + c->nLanes - 1) % c->nLanes; }   // lane of the frame rendered last
void *target_ptr(RtContext *c, int which, int &channels) {
    const int l = last_lane(c);
    switch (which) {
        case RT_TARGET_COLOR: channels = 4; return c->dColor[l];
        case RT_TARGET_MOTION: channels = 2; return c->dMotion[l];
        case RT_TARGET_GPOS: channels = 4; return c->dGPos[l];
        case RT_TARGET_GNRM: channels = 4; return c->dGNrm[l];
        default: channels = 0; return nullptr;
    }
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// stage timing helpers (used by rt_wave.hip through RtStageTimer)
void rt_stage_begin(RtContext *c, int stage, hipStream_t on) {
    if (!c->timing) return;
    StageEvent ev;
    ev.stage = stage;
    if (!c->freeEvents.empty()) { ev.a = c->freeEvents.back().first; ev.b = c->freeEvents.back().second; c->freeEvents.pop_back(); }
    else { (void)hipEventCreate(&ev.a); (void)hipEventCreate(&ev.b); }
    (void)hipEventRecord(ev.a, on ? on : c->stream);
    c->pending.push_back(ev);
}
void rt_stage_end(RtContext *c, int stage, int launches, hipStream_t on) {
    if (!c->timing) return;
    for (size_t i = c->pending.size(); i-- > 0;)
        if (c->pending[i].stage == stage) { (void)hipEventRecord(c->pending[i].b, on ? on : c->stream); break; }
    c->stageLaunches[stage] += (uint64_t)launches;
}
// Long timed runs: fold the events that have completed into the totals so that the pending list stays short (no host sync).
static void harvest_stage_events(RtContext *c) {
    if (c->pending.size() < 1024) return;
    size_t done = 0;
    while (done < c->pending.size() && hipEventQuery(c->pending[done].b) == hipSuccess) {
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, c->pending[done].a, c->pending[done].b) == hipSuccess) c->stageMs[c->pending[done].stage] += ms;
        c->freeEvents.emplace_back(c->pending[done].a, c->pending[done].b);
        ++done;
    }
    c->pending.erase(c->pending.begin(), c->pending.begin() + (std::ptrdiff_t)done);
}
static void resolve_stage_events(RtContext *c) {
    (void)sync_all(c);
    for (auto &ev : c->pending) {
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, ev.a, ev.b) == hipSuccess) c->stageMs[ev.stage] += ms;
        c->freeEvents.emplace_back(ev.a, ev.b);
    }
    c->pending.clear();
}

extern "C" {

const char *rt_stage_name(int stage) { return (stage >= 0 && stage < RT_MAX_STAGES) ? kStageNames[stage] : ""; }

const char *rt_last_error(const RtContext *ctx) { return ctx ? ctx->err.c_str() : g_createError.c_str(); }

int rt_create(const RtDeviceConfig *cfg, RtContext **out) {
    if (!cfg || !out) return fail(nullptr, RT_ERR_INVALID, "rt_create: null argument");
    *out = nullptr;
    if (cfg->worldSize < 1 || cfg->rank < 0 || cfg->rank >= cfg->worldSize) return fail(nullptr, RT_ERR_INVALID, "rt_create: bad rank/worldSize");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(nullptr, RT_ERR_NO_DEVICE, "rt_create: no HIP device (%s); this library has no CPU path", hipGetErrorString(e));
    if (cfg->device < 0 || cfg->device >= ndev) return fail(nullptr, RT_ERR_INVALID, "rt_create: device %d of %d", cfg->device, ndev);
    e = hipSetDevice(cfg->device);
    if (e != hipSuccess) return fail(nullptr, RT_ERR_NO_DEVICE, "hipSetDevice: %s", hipGetErrorString(e));
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, cfg->device);
    if (e != hipSuccess) return fail(nullptr, RT_ERR_HIP, "hipGetDeviceProperties: %s", hipGetErrorString(e));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(nullptr, RT_ERR_UNSUPPORTED, "rt_create: device is %s, this library carries gfx950 code only", prop.gcnArchName);
    RtContext *c = new RtContext();
    c->cfg = *cfg;
    // measured on MI355X (1080p / 4 spp): whole frame on one GPU 3.0 / 2.48 / 2.44 / 2.58 ms with 1 / 2 / 3 / 4 lanes; one rank of
    // eight (1/8 of the tiles, latency-bound stages) 0.92 / 0.64 / 0.56 / 0.51 ms
    // round 3, batches of eight frames with 75 % persistent grids: 1.76 / 1.75 ms per frame with 3 / 4 lanes
    c->nLanes = 4;
    if (const char *e = getenv("RT_LANES")) c->nLanes = std::max(1, std::min(RT_MAX_LANES, atoi(e)));
    // RT_LANES=1 = one frame (launch set) in flight.  The COLOR0 ring still needs two buffers -- with one, a moving frame's reprojection would read
    // history texels its own resolve is overwriting (rt_taa.glsl:128 reads the previous frame at arbitrary pixels) -- so it is two lanes whose
    // frames wait for their predecessor from the first kernel on.
    if (c->nLanes == 1) { c->nLanes = 2; c->serialFrames = true; }
    bool ok = hipMalloc(&c->dCounters, 16 * sizeof(unsigned long long)) == hipSuccess;
    // EXPERIMENT RT_CU_SPLIT=k (rt_wave.hip): the lanes' own streams -- the traversal launches -- keep 8 - k eighths of the CUs
    uint32_t cuMask[8];
    bool masked = false;
    if (const char *e = getenv("RT_CU_SPLIT")) {
        const int k = std::max(1, std::min(7, atoi(e)));
        for (int i = 0; i < 8; ++i) { uint32_t m = 0; for (int b = 0; b < 32; ++b) if (((i * 32 + b) & 7) >= k) m |= 1u << b; cuMask[i] = m; }
        masked = true;
    }
    for (int i = 0; ok && i < c->nLanes; ++i)
        ok = (masked ? hipExtStreamCreateWithCUMask(&c->lanes[i], 8, cuMask) : hipStreamCreateWithFlags(&c->lanes[i], hipStreamNonBlocking)) == hipSuccess &&
             hipMalloc(&c->dFrame[i], sizeof(DevFrame)) == hipSuccess && hipEventCreateWithFlags(&c->evDone[i], hipEventDisableTiming) == hipSuccess;
    if (!ok) {
        rt_destroy(c);
        return fail(nullptr, RT_ERR_HIP, "rt_create: stream/alloc failed");
    }
    c->stream = c->lanes[0];
    (void)hipMemset(c->dCounters, 0, 16 * sizeof(unsigned long long));
    c->cus = prop.multiProcessorCount;
    // ray-queue arenas: RT_ARENAS of them shared by the lanes (default below; = lanes: one each, as in rounds 1-3)
    int arenas = std::min(c->nLanes, kDefaultArenas);
    if (const char *e = getenv("RT_ARENAS")) arenas = std::max(1, std::min(atoi(e), c->nLanes));
    c->arenaPool = rt_arena_pool_create(arenas);
    for (int i = 0; i < c->nLanes; ++i) c->wave[i] = rt_wave_create(prop.multiProcessorCount, c->arenaPool, i);
    c->lastStream = c->stream;
    int rc = rt_upload_env(c, nullptr, 0, 0);   // dummy cube map like Application::initState (application.cpp:281)
    if (rc != RT_OK) { g_createError = c->err; rt_destroy(c); return rc; }
    *out = c;
    return RT_OK;
}

void rt_destroy(RtContext *c) {
    if (!c) return;
    (void)hipSetDevice(c->cfg.device);
    (void)sync_all(c);
    (void)rt_comm_destroy(c);
    free_targets(c);
    for (int i = 0; i < RT_MAX_LANES; ++i) { if (c->hybrid[i]) rt_hybrid_destroy(c->hybrid[i]); if (c->wave[i]) rt_wave_destroy(c->wave[i]); if (c->dFrame[i]) (void)hipFree(c->dFrame[i]); if (c->evDone[i]) (void)hipEventDestroy(c->evDone[i]); }
    rt_arena_pool_destroy(c->arenaPool);
    for (int i = 0; i < RT_MAX_LANES; ++i) if (c->lanes[i]) (void)hipStreamDestroy(c->lanes[i]);   // c->stream is lanes[0]
    if (c->dWNodes) (void)hipFree(c->dWNodes);
    if (c->dW4) (void)hipFree(c->dW4);
    if (c->dQ4) (void)hipFree(c->dQ4);
    if (c->dLeafBox) (void)hipFree(c->dLeafBox);
    if (c->dWNodesW) (void)hipFree(c->dWNodesW);
    if (c->dWF) (void)hipFree(c->dWF);
    if (c->dIN2) (void)hipFree(c->dIN2);
    if (c->dIPairs) (void)hipFree(c->dIPairs);
    if (c->dIN4) (void)hipFree(c->dIN4);
    if (c->dIQ4) (void)hipFree(c->dIQ4);
    if (c->dILeafBox) (void)hipFree(c->dILeafBox);
    if (c->dPairs) (void)hipFree(c->dPairs);
    if (c->dTris) (void)hipFree(c->dTris);
    if (c->dEnv) (void)hipFree(c->dEnv);
    if (c->dCounters) (void)hipFree(c->dCounters);
    if (c->dStaging) (void)hipFree(c->dStaging);
    for (auto &ev : c->pending) { (void)hipEventDestroy(ev.a); (void)hipEventDestroy(ev.b); }
    for (auto &p : c->freeEvents) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
    delete c;
}

int rt_upload_bvh(RtContext *c, const float *nodes12, int nNodes, const float *tris12, int nTris) {
    if (!c) return RT_ERR_INVALID;
    if (nNodes < 0 || nTris < 0 || (nNodes > 0 && !nodes12) || (nTris > 0 && !tris12)) return fail(c, RT_ERR_INVALID, "rt_upload_bvh: bad arguments");
    return guarded(c, "rt_upload_bvh", [&]() -> int {
    (void)hipSetDevice(c->cfg.device);
    HIP_TRY(c, sync_all(c));
    if (c->dWNodes) (void)hipFree(c->dWNodes);
    if (c->dW4) (void)hipFree(c->dW4);
    if (c->dQ4) (void)hipFree(c->dQ4);
    if (c->dLeafBox) (void)hipFree(c->dLeafBox);
    if (c->dWNodesW) (void)hipFree(c->dWNodesW);
    if (c->dWF) (void)hipFree(c->dWF);
    if (c->dIN2) (void)hipFree(c->dIN2);
    if (c->dIPairs) (void)hipFree(c->dIPairs);
    if (c->dIN4) (void)hipFree(c->dIN4);
    if (c->dIQ4) (void)hipFree(c->dIQ4);
    if (c->dILeafBox) (void)hipFree(c->dILeafBox);
    c->dIN2 = c->dIPairs = c->dIN4 = c->dIQ4 = c->dILeafBox = nullptr; c->implD = c->implR = 0;
    if (c->dPairs) (void)hipFree(c->dPairs);
    if (c->dTris) (void)hipFree(c->dTris);
    c->dWNodes = c->dW4 = c->dTris = c->dWNodesW = c->dPairs = c->dQ4 = c->dLeafBox = c->dWF = nullptr;
    c->nNodes = c->nTris = c->nInner = 0;
    c->treeDepth = 0;
    c->nWide4 = c->nPairs = c->nFused = 0;
    c->sceneFlags = 0;
    c->anyStack = 0;
    if (nNodes == 0 || nTris == 0) return RT_OK;
    if (nTris >= (1 << 28)) return fail(c, RT_ERR_UNSUPPORTED, "rt_upload_bvh: %d triangles exceed the 2^28 leaf encoding", nTris);

    // Decode the reference's float-encoded links exactly as nodeFetch does (rt_bvh.glsl:97-100).
    struct N { int left, right, first, count; };
    std::vector<N> nd((size_t)nNodes);
    std::vector<int> innerIdx((size_t)nNodes, -1);
    int nInner = 0;
    for (int i = 0; i < nNodes; ++i) {
        const float *p = nodes12 + (size_t)i * 12;
        nd[(size_t)i] = {(int)(p[3] + 0.5f), (int)(p[7] + 0.5f), (int)(p[8] + 0.5f), (int)(p[9] + 0.5f)};
        const N &n = nd[(size_t)i];
        if (n.count > 0) {
            if (n.count > 8 || n.first < 0 || n.first + n.count > nTris)
                return fail(c, RT_ERR_UNSUPPORTED, "rt_upload_bvh: leaf %d has first=%d count=%d (leaves hold 1..8 triangles)", i, n.first, n.count);
        } else {
            if (n.left <= 0 || n.right <= 0 || n.left >= nNodes || n.right >= nNodes)
                return fail(c, RT_ERR_INVALID, "rt_upload_bvh: inner node %d has children %d,%d", i, n.left, n.right);
            innerIdx[(size_t)i] = nInner++;
        }
    }
    auto refOf = [&](int node) {
        const N &n = nd[(size_t)node];
        return (n.count > 0) ? -(((n.first << 3) | (n.count - 1)) + 1) : innerIdx[(size_t)node];
    };
    // Triangle PAIR records for the wavefront pipeline's traversal kernels: what bounds them is the number of 16-byte gather
    // loads per ray, and a 48-byte triangle record carries only 36 bytes of payload.  Two triangles of a leaf are
    // packed into 80 bytes = 5 loads instead of 6: floats [0..8] = v0,e1,e2 of the first, [9..17] of the second, [18] = index of the
    // first in the reference's triangle array (closest-hit results name triangles by it), [19] unused.  A leaf owns
    // ceil(count/2) consecutive records; its reference in the wavefront node arrays addresses the first of them.
    std::vector<float> pairs;
    std::vector<int> pairRefOf((size_t)nNodes, 0);
    for (int i = 0; i < nNodes; ++i) {
        const N &n = nd[(size_t)i];
        if (n.count <= 0) continue;
        const size_t rec = pairs.size() / 20;
        if (rec + 8 >= ((size_t)1 << 28)) return fail(c, RT_ERR_UNSUPPORTED, "rt_upload_bvh: pair records exceed the 2^28 leaf encoding");
        pairRefOf[(size_t)i] = -((int)((rec << 3) | (size_t)(n.count - 1)) + 1);
        for (int t = 0; t < n.count; t += 2) {
            float r[20] = {0};
            for (int h = 0; h < 2 && t + h < n.count; ++h) {
                const float *q = tris12 + (size_t)(n.first + t + h) * 12;
                const float nine[9] = {q[0], q[1], q[2], q[4], q[5], q[6], q[8], q[9], q[10]};
                std::memcpy(&r[9 * h], nine, sizeof nine);
            }
            const uint32_t orig = (uint32_t)(n.first + t);
            std::memcpy(&r[18], &orig, 4);
            if (t + 1 >= n.count) std::memcpy(&r[9], &orig, 4);   // a record with one triangle: the index again in the unused tenth float (3-load fetch)
            pairs.insert(pairs.end(), r, r + 20);
        }
    }
    pairs.resize(pairs.size() + 8 * 20, 0.0f);   // groups are fetched without a bounds branch
    auto refOfW = [&](int node) { return (nd[(size_t)node].count > 0) ? pairRefOf[(size_t)node] : innerIdx[(size_t)node]; };
    std::vector<float> wn((size_t)std::max(nInner, 1) * 16, 0.0f);
    for (int i = 0; i < nNodes; ++i) {
        if (innerIdx[(size_t)i] < 0) continue;
        const float *L = nodes12 + (size_t)nd[(size_t)i].left * 12, *R = nodes12 + (size_t)nd[(size_t)i].right * 12;
        float *o = &wn[(size_t)innerIdx[(size_t)i] * 16];
        int rl = refOf(nd[(size_t)i].left), rr = refOf(nd[(size_t)i].right);
        o[0] = L[0]; o[1] = L[1]; o[2] = L[2]; std::memcpy(&o[3], &rl, 4);
        o[4] = L[4]; o[5] = L[5]; o[6] = L[6]; std::memcpy(&o[7], &rr, 4);
        o[8] = R[0]; o[9] = R[1]; o[10] = R[2];
        o[12] = R[4]; o[13] = R[5]; o[14] = R[6];
    }
    std::vector<float> wnW = wn;   // the same records with pair-record leaf references (wavefront closest-hit kernels)
    for (int i = 0; i < nNodes; ++i) {
        if (innerIdx[(size_t)i] < 0) continue;
        float *o = &wnW[(size_t)innerIdx[(size_t)i] * 16];
        int rl = refOfW(nd[(size_t)i].left), rr = refOfW(nd[(size_t)i].right);
        std::memcpy(&o[3], &rl, 4);
        std::memcpy(&o[7], &rr, 4);
    }
    // depth of the tree (iterative), bounds the traversal stack: one deferred sibling per level
    int depth = 0;
    {
        std::vector<std::pair<int, int>> st{{0, 1}};
        size_t visited = 0;
        while (!st.empty()) {
            auto [n, d] = st.back();
            st.pop_back();
            if (++visited > (size_t)nNodes) return fail(c, RT_ERR_INVALID, "rt_upload_bvh: node links form a cycle");
            depth = std::max(depth, d);
            if (nd[(size_t)n].count <= 0) { st.push_back({nd[(size_t)n].left, d + 1}); st.push_back({nd[(size_t)n].right, d + 1}); }
        }
    }
    // 4-wide nodes for any-hit rays: every binary inner node at an even level absorbs its inner children, so one
    // 128-byte record holds up to four grandchild boxes -- stored component-wise, so that the 28 payload floats take 7 of the
    // record's 8 sixteen-byte pieces and a visit costs 7 gather loads.  A child box is only skipped (never tested) when it is an
    // intermediate node; by monotonicity of the slab arithmetic a grandchild that passes its own test also passes
    // its parent's, so the set of triangles tested -- and hence every any-hit answer -- is unchanged.
    std::vector<float> w4;
    int rootRef4 = refOfW(0);
    if (nd[0].count <= 0) {
        struct Job { int bin; size_t at; };   // fill node `at` (index into w4 / 32) from binary node `bin`
        std::vector<Job> jobs;
        w4.resize(32, 0.0f);
        jobs.push_back({0, 0});
        rootRef4 = 0;
        while (!jobs.empty()) {
            Job jb = jobs.back();
            jobs.pop_back();
            int kids[4], nk = 0;
            for (int ch : {nd[(size_t)jb.bin].left, nd[(size_t)jb.bin].right}) {
                if (nd[(size_t)ch].count > 0) kids[nk++] = ch;
                else { kids[nk++] = nd[(size_t)ch].left; kids[nk++] = nd[(size_t)ch].right; }
            }
            for (int i = 0; i < 4; ++i) {
                int ref = RT_NO_CHILD;
                if (i < nk) {
                    const float *b = nodes12 + (size_t)kids[i] * 12;
                    if (nd[(size_t)kids[i]].count > 0) ref = refOfW(kids[i]);
                    else {
                        ref = (int)(w4.size() / 32);
                        w4.resize(w4.size() + 32, 0.0f);
                        jobs.push_back({kids[i], (size_t)ref});
                    }
                    float *o = &w4[jb.at * 32];     // SoA: [min.x x4][min.y x4][min.z x4][max.x x4][max.y x4][max.z x4][ref x4][-]
                    o[0 + i] = b[0]; o[4 + i] = b[1]; o[8 + i] = b[2];
                    o[12 + i] = b[4]; o[16 + i] = b[5]; o[20 + i] = b[6];
                } else {
                    // absent child: NaN box -- (NaN - ro) * rdInv = NaN, v_min/v_max drop NaN operands, tmax = NaN, and
                    // "tmax >= tmin" is false, so the traversal kernel needs no child-present branch
                    float *o = &w4[jb.at * 32];
                    const float qnan = std::nanf("");
                    o[0 + i] = o[4 + i] = o[8 + i] = o[12 + i] = o[16 + i] = o[20 + i] = qnan;
                }
                std::memcpy(&w4[jb.at * 32 + 24 + (size_t)i], &ref, 4);
            }
        }
    } else w4.resize(32, 0.0f);
    // EXPERIMENT (RT_ANYHIT_TREE=sah): any-hit answers do not depend on the tree, only on which reference leaves pass their own
    // exact box test -- so the 4-wide tree may be ANY tree over the reference's leaves whose inner boxes contain their leaves' boxes.  Binned SAH
    // (16 bins, cost = area x triangles) over the leaves, collapsed to four children by pulling up the child of largest area.  Off by default: it
    // needs more node visits than the collapse of the balanced median-split tree on the bench mesh.
    int anyStack = 0;
    if (const char *e = getenv("RT_ANYHIT_TREE")) if (std::string(e) == "sah" && nd[0].count <= 0) {
        struct Leaf { float lo[3], hi[3], cen[3]; int ref; float w; };
        std::vector<Leaf> leaves;
        for (int i = 0; i < nNodes; ++i) if (nd[(size_t)i].count > 0) {
            Leaf L;
            const float *b = nodes12 + (size_t)i * 12;
            for (int a = 0; a < 3; ++a) { L.lo[a] = b[a]; L.hi[a] = b[4 + a]; L.cen[a] = 0.5f * (b[a] + b[4 + a]); }
            L.ref = refOfW(i); L.w = (float)nd[(size_t)i].count;
            leaves.push_back(L);
        }
        struct BNode { float lo[3], hi[3]; int left, right, leaf; };
        std::vector<BNode> B;
        std::vector<int> ids(leaves.size());
        for (size_t i = 0; i < ids.size(); ++i) ids[i] = (int)i;
        auto area = [](const float *lo, const float *hi) { float ex = std::max(hi[0] - lo[0], 0.0f), ey = std::max(hi[1] - lo[1], 0.0f), ez = std::max(hi[2] - lo[2], 0.0f); return 2.0f * (ex * ey + ey * ez + ez * ex); };
        struct Job { int begin, end, node, depth; };
        std::vector<Job> todo;
        B.push_back(BNode{});
        todo.push_back({0, (int)ids.size(), 0, 0});
        constexpr int NB = 16;
        while (!todo.empty()) {
            const Job jb = todo.back();
            todo.pop_back();
            BNode bn{};
            for (int a = 0; a < 3; ++a) { bn.lo[a] = 3.0e38f; bn.hi[a] = -3.0e38f; }
            float clo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, chi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
            for (int i = jb.begin; i < jb.end; ++i) {
                const Leaf &L = leaves[(size_t)ids[(size_t)i]];
                for (int a = 0; a < 3; ++a) { bn.lo[a] = std::min(bn.lo[a], L.lo[a]); bn.hi[a] = std::max(bn.hi[a], L.hi[a]); clo[a] = std::min(clo[a], L.cen[a]); chi[a] = std::max(chi[a], L.cen[a]); }
            }
            bn.left = bn.right = -1; bn.leaf = -1;
            if (jb.end - jb.begin == 1) { bn.leaf = ids[(size_t)jb.begin]; B[(size_t)jb.node] = bn; continue; }
            int bestAxis = -1, bestSplit = -1;
            float bestCost = 3.0e38f;
            for (int a = 0; a < 3 && jb.depth < 40; ++a) {
                const float ext = chi[a] - clo[a];
                if (!(ext > 0.0f)) continue;
                float blo[NB][3], bhi[NB][3], bw[NB];
                for (int k = 0; k < NB; ++k) { bw[k] = 0.0f; for (int q = 0; q < 3; ++q) { blo[k][q] = 3.0e38f; bhi[k][q] = -3.0e38f; } }
                for (int i = jb.begin; i < jb.end; ++i) {
                    const Leaf &L = leaves[(size_t)ids[(size_t)i]];
                    const int k = std::min(NB - 1, (int)((L.cen[a] - clo[a]) / ext * NB));
                    bw[k] += L.w;
                    for (int q = 0; q < 3; ++q) { blo[k][q] = std::min(blo[k][q], L.lo[q]); bhi[k][q] = std::max(bhi[k][q], L.hi[q]); }
                }
                float rlo[NB][3], rhi[NB][3], rw[NB];
                float alo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, ahi[3] = {-3.0e38f, -3.0e38f, -3.0e38f}, aw = 0.0f;
                for (int k = NB - 1; k >= 0; --k) {
                    aw += bw[k];
                    for (int q = 0; q < 3; ++q) { alo[q] = std::min(alo[q], blo[k][q]); ahi[q] = std::max(ahi[q], bhi[k][q]); rlo[k][q] = alo[q]; rhi[k][q] = ahi[q]; }
                    rw[k] = aw;
                }
                float llo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, lhi[3] = {-3.0e38f, -3.0e38f, -3.0e38f}, lw = 0.0f;
                for (int k = 0; k + 1 < NB; ++k) {
                    lw += bw[k];
                    for (int q = 0; q < 3; ++q) { llo[q] = std::min(llo[q], blo[k][q]); lhi[q] = std::max(lhi[q], bhi[k][q]); }
                    if (lw == 0.0f || rw[k + 1] == 0.0f) continue;
                    const float cost = area(llo, lhi) * lw + area(rlo[k + 1], rhi[k + 1]) * rw[k + 1];
                    if (cost < bestCost) { bestCost = cost; bestAxis = a; bestSplit = k; }
                }
            }
            int mid;
            if (bestAxis < 0) {   // degenerate centroids or the depth cap: median by index
                mid = (jb.begin + jb.end) / 2;
            } else {
                const float ext = chi[bestAxis] - clo[bestAxis];
                auto it = std::partition(ids.begin() + jb.begin, ids.begin() + jb.end, [&](int id) {
                    return std::min(NB - 1, (int)((leaves[(size_t)id].cen[bestAxis] - clo[bestAxis]) / ext * NB)) <= bestSplit; });
                mid = (int)(it - ids.begin());
                if (mid == jb.begin || mid == jb.end) mid = (jb.begin + jb.end) / 2;
            }
            bn.left = (int)B.size(); B.push_back(BNode{});
            bn.right = (int)B.size(); B.push_back(BNode{});
            B[(size_t)jb.node] = bn;
            todo.push_back({jb.begin, mid, bn.left, jb.depth + 1});
            todo.push_back({mid, jb.end, bn.right, jb.depth + 1});
        }
        // collapse to four children
        std::vector<float> s4(32, 0.0f);
        struct J4 { int bin; size_t at; int depth; };
        std::vector<J4> jobs4{{0, 0, 1}};
        int depth4 = 1;
        while (!jobs4.empty()) {
            const J4 jb = jobs4.back();
            jobs4.pop_back();
            depth4 = std::max(depth4, jb.depth);
            std::vector<int> kids{B[(size_t)jb.bin].left, B[(size_t)jb.bin].right};
            while (kids.size() < 4) {
                int pick = -1;
                float best = -1.0f;
                for (size_t i = 0; i < kids.size(); ++i) if (B[(size_t)kids[i]].leaf < 0) { const float ar = area(B[(size_t)kids[i]].lo, B[(size_t)kids[i]].hi); if (ar > best) { best = ar; pick = (int)i; } }
                if (pick < 0) break;
                const int k = kids[(size_t)pick];
                kids.erase(kids.begin() + pick);
                kids.push_back(B[(size_t)k].left); kids.push_back(B[(size_t)k].right);
            }
            for (int i = 0; i < 4; ++i) {
                int ref = RT_NO_CHILD;
                float *o = &s4[jb.at * 32];
                if (i < (int)kids.size()) {
                    const BNode &kb = B[(size_t)kids[(size_t)i]];
                    if (kb.leaf >= 0) ref = leaves[(size_t)kb.leaf].ref;
                    else { ref = (int)(s4.size() / 32); s4.resize(s4.size() + 32, 0.0f); jobs4.push_back({kids[(size_t)i], (size_t)ref, jb.depth + 1}); o = &s4[jb.at * 32]; }
                    // a leaf child carries the reference's exact leaf box (the test that decides which triangles are tested); an inner child the union of its leaves' boxes
                    const float *lo = kb.leaf >= 0 ? leaves[(size_t)kb.leaf].lo : kb.lo, *hi = kb.leaf >= 0 ? leaves[(size_t)kb.leaf].hi : kb.hi;
                    o[0 + i] = lo[0]; o[4 + i] = lo[1]; o[8 + i] = lo[2]; o[12 + i] = hi[0]; o[16 + i] = hi[1]; o[20 + i] = hi[2];
                } else {
                    const float qnan = std::nanf("");
                    o[0 + i] = o[4 + i] = o[8 + i] = o[12 + i] = o[16 + i] = o[20 + i] = qnan;
                }
                std::memcpy(&s4[jb.at * 32 + 24 + (size_t)i], &ref, 4);
            }
        }
        if (3 * depth4 <= 60) { w4.swap(s4); rootRef4 = 0; anyStack = 3 * depth4; }
    }
    // Exact stack need of the any-hit walk (round 4): a visit of a node with nc children pushes at most nc - 1 entries (one child is gone on with),
    // so S(node) = nc - 1 + max over its inner children S(child).  Round 3 sized the stack as 3 per two binary levels INCLUDING the leaf level: 24 entries
    // for the bench mesh, where 21 are enough -- and 21 x 4 B x 256 threads let seven workgroups share a CU's 160 KB of LDS instead of six.
    if (rootRef4 == 0 && w4.size() >= 32) {
        const size_t n4 = w4.size() / 32;
        std::vector<int> need(n4, -1);
        std::vector<std::pair<size_t, int>> st{{0, 0}};   // (node, next child to look at)
        while (!st.empty()) {
            auto &[nn, ci] = st.back();
            if (ci < 4) {
                int ref;
                std::memcpy(&ref, &w4[nn * 32 + 24 + (size_t)ci], 4);
                ++ci;
                if (ref >= 0 && ref != RT_NO_CHILD && (size_t)ref < n4 && need[(size_t)ref] < 0) st.push_back({(size_t)ref, 0});
                continue;
            }
            int nc = 0, deepest = 0;
            for (int i = 0; i < 4; ++i) {
                int ref;
                std::memcpy(&ref, &w4[nn * 32 + 24 + (size_t)i], 4);
                if (ref == RT_NO_CHILD) continue;
                ++nc;
                if (ref >= 0 && (size_t)ref < n4) deepest = std::max(deepest, need[(size_t)ref]);
            }
            need[nn] = std::max(nc - 1, 0) + deepest;
            st.pop_back();
        }
        anyStack = std::max(need[0], 1);
    }
    // Round 4: the any-hit tree once more with QUANTISED child boxes -- 64 bytes per 4-wide node instead of 112, four gather loads per
    // visit instead of seven.  Only the reference's LEAVES need their exact boxes (an any-hit answer is the OR over the leaves that pass their own box test);
    // an inner box may be any box that contains them, and the slab arithmetic is monotonic in the box, so a decoded box that is checked HERE, in the very
    // float expression the kernel decodes with (fmaf(q, 2^e, origin)), to contain the child's box passes whenever the child's does.  A leaf child passes
    // the quantised test first and its exact box -- kept in `leafBox`, indexed from the leaf's first pair record (below) -- in the leaf phase.
    // Built (and walked, rt_wave.hip launch_trace) when the 112-byte nodes outgrow one XCD's 4 MB L2 (kQNodesAbove): the decode costs 48 VALU operations per
    // visit and a wave per SIMD, which a cache-resident tree does not earn back -- any-hit launch per frame, exact / quantised nodes, one launch set in flight:
    // 20 k triangles 0.57 / 0.63-0.66 ms, 82 k (bench mesh, 0.6 MB of nodes) 0.66 / 0.74, 328 k (2.4 MB) 0.81 / 0.83-0.84, 1 M (9.8 MB) 15.9 / 13.6
    // (profiles/r04_experiments.txt 19).  RT_QNODES=0 / 2 forces either.
    //   piece 0: origin.xyz (float), biased exponents ex | ey << 8 | ez << 16       piece 1: lo.x lo.y lo.z hi.x, one byte per child
    //   piece 2: hi.y hi.z - -                                                       piece 3: the four child references of the 112-byte node
    std::vector<uint32_t> q4;
    std::vector<float> leafBox;
    const int qmode = getenv("RT_QNODES") ? atoi(getenv("RT_QNODES")) : -1;
    if (rootRef4 == 0 && (qmode > 0 || (qmode < 0 && (w4.size() / 32) * 112 > kQNodesAbove))) {
        const size_t n4 = w4.size() / 32;
        q4.assign(n4 * 16, 0u);
        bool okQ = true;
        for (size_t nn = 0; nn < n4 && okQ; ++nn) {
            const float *o = &w4[nn * 32];
            int refs[4];
            std::memcpy(refs, o + 24, 16);
            uint32_t *q = &q4[nn * 16];
            float org[3], scale[3];
            uint32_t exps = 0;
            for (int a = 0; a < 3; ++a) {
                float lo = INFINITY, hi = -INFINITY;
                for (int i = 0; i < 4; ++i) if (refs[i] != RT_NO_CHILD) { lo = std::min(lo, o[4 * a + i]); hi = std::max(hi, o[12 + 4 * a + i]); }
                if (!(lo <= hi)) { lo = hi = 0.0f; }
                int eb = 1;
                const double ext = ((double)hi - (double)lo) / 255.0;
                if (ext > 0.0) { int e2; (void)std::frexp(ext, &e2); eb = std::max(1, e2 - 1 + 127); }
                while (eb <= 254 && std::fmaf(255.0f, std::ldexp(1.0f, eb - 127), lo) < hi) ++eb;
                if (eb > 254) { okQ = false; break; }
                org[a] = lo; scale[a] = std::ldexp(1.0f, eb - 127);
                exps |= (uint32_t)eb << (8 * a);
                std::memcpy(&q[a], &lo, 4);
            }
            if (!okQ) break;
            q[3] = exps;
            for (int i = 0; i < 4; ++i) {
                q[12 + i] = (uint32_t)refs[i];
                if (refs[i] == RT_NO_CHILD) continue;
                for (int a = 0; a < 3; ++a) {
                    const float lo = o[4 * a + i], hi = o[12 + 4 * a + i];
                    int ql = (int)std::floor(((double)lo - (double)org[a]) / (double)scale[a]);
                    ql = std::max(0, std::min(255, ql));
                    while (ql > 0 && std::fmaf((float)ql, scale[a], org[a]) > lo) --ql;
                    int qh = (int)std::ceil(((double)hi - (double)org[a]) / (double)scale[a]);
                    qh = std::max(0, std::min(255, qh));
                    while (qh < 255 && std::fmaf((float)qh, scale[a], org[a]) < hi) ++qh;
                    if (std::fmaf((float)ql, scale[a], org[a]) > lo || std::fmaf((float)qh, scale[a], org[a]) < hi) okQ = false;
                    const int wl = 4 + a, wh = a == 0 ? 7 : 7 + a;      // words: lo.x lo.y lo.z hi.x | hi.y hi.z
                    q[wl] |= (uint32_t)ql << (8 * i);
                    q[wh] |= (uint32_t)qh << (8 * i);
                }
            }
        }
        if (okQ) {
            // A leaf's box sits at index first / R, R = the smallest number of pair records any leaf owns: consecutive leaves are at least R records apart,
            // so the quotient is distinct per leaf, and the array is dense when the leaves are alike (a median-split tree: record counts differ by at most one).
            // The kernel divides by multiplying with ceil(2^32 / R) (exact for first < 2^28, R <= 8); R = 1: the identity (magic 0).
            int rmin = 8;
            for (int i = 0; i < nNodes; ++i) if (nd[(size_t)i].count > 0) rmin = std::min(rmin, (nd[(size_t)i].count + 1) / 2);
            if (getenv("RT_QNODES_SPARSE_BOXES")) rmin = 1;      // EXPERIMENT: one slot per pair record, as first built
            c->leafBoxMagic = rmin <= 1 ? 0u : (uint32_t)((((uint64_t)1 << 32) + (uint64_t)rmin - 1) / (uint64_t)rmin);
            leafBox.assign(((pairs.size() / 20) / (size_t)std::max(rmin, 1) + 1) * 8, 0.0f);
            std::vector<char> taken(leafBox.size() / 8, 0);   // (host only: nothing but boxes is uploaded)
            c->nLeafBoxes = 0;
            for (int i = 0; i < nNodes; ++i) if (nd[(size_t)i].count > 0) {
                ++c->nLeafBoxes;
                const size_t first = (size_t)(-pairRefOf[(size_t)i] - 1) >> 3;
                const size_t at = c->leafBoxMagic ? (size_t)(((uint64_t)first * c->leafBoxMagic) >> 32) : first;
                const float *b = nodes12 + (size_t)i * 12;
                const float box[8] = {b[0], b[1], b[2], b[4], b[5], b[6], 0.0f, 0.0f};
                if (at * 8 + 8 > leafBox.size() || taken[at]) { okQ = false; break; }   // (cannot happen: distinct quotients)
                std::memcpy(&leafBox[at * 8], box, sizeof box);
                taken[at] = 1;
            }
            if (!okQ) { q4.clear(); leafBox.clear(); }
        } else q4.clear();
        if (!okQ) {   // asked for and not built: say so instead of falling back silently
            c->sceneFlags |= RT_SCENE_QNODES_REJECTED;
            if (getenv("RT_VERBOSE")) fprintf(stderr, "[rt_upload_bvh] quantised any-hit nodes rejected (exponent range or leaf-box index collision): walking the exact 112-byte nodes\n");
        }
    }
    // Round 5 -- fused closest-hit records.  The closest-hit walk must keep the reference's visiting order (rt_bvh.glsl:205-241: near child first, far
    // child behind the pop-time cull; ties and triangles a rounding in front of their leaf box make the answer depend on it), but nothing says it
    // must spend one dependent memory round trip per binary node.  Whenever the walk enters a child X of a node N it visits X next, and X's record (the
    // boxes of X's children) is known as soon as N's index is -- so the records of N's two children are stored TOGETHER under N's index, 128 bytes = one
    // cache line per even-level inner node N ("hub"):
    //     pieces 0-3: [A1.min, ref A1] [A1.max, ref A2] [A2.min, -] [A2.max, -]      A = N.left,  A1 / A2 = A's children
    //     pieces 4-7: the same for B = N.right
    // A leaf child X is stored as a half with ONE box: [X.min, leaf ref] [X.max, RT_NO_CHILD] [NaN box].  References are hub indices (>= 0) or the pair-record
    // leaf codes of wnodesW (< 0).  The kernel (k_trace<.., FUSE>) fetches the eight pieces in one round trip and makes the reference's two steps from them:
    // the step at N needs the boxes of A and B themselves, which are not stored -- they are the unions of their children's boxes, and the slab values of a
    // union follow from the children's by min / max (per axis: tsm = min(tsm1, tsm2), tbg = max(tbg1, tbg2); tests/test_fused_nodes.py pins that identity
    // against the slab test of the union box, signed zeros, infinities and the NaN of 0 * inf included).  That needs every inner box to BE the union of its children's, bit for bit --
    // true of the reference's builder (bounds over a range = the union of the bounds over its halves, bvh.cpp:41-60) and of rt_build_bvh_gpu, checked
    // here for whatever was uploaded; a tree that fails the check keeps the 64-byte records (RT_SCENE_NOT_FUSED in RtSceneInfo.flags).
    std::vector<float> wF;
    {
        // built only where the option is on (both forms are measured options, off by default: no second and third copy of the tree in device memory otherwise)
        const bool wantF = getenv("RT_FUSED") && atoi(getenv("RT_FUSED")) != 0;
        bool okF = wantF && nd[0].count <= 0;
        for (int i = 0; i < nNodes && okF; ++i) {
            if (nd[(size_t)i].count > 0) continue;
            const float *P = nodes12 + (size_t)i * 12, *L = nodes12 + (size_t)nd[(size_t)i].left * 12, *R = nodes12 + (size_t)nd[(size_t)i].right * 12;
            for (int a = 0; a < 3; ++a)
                if (!(P[a] == std::min(L[a], R[a])) || !(P[4 + a] == std::max(L[4 + a], R[4 + a]))) okF = false;
        }
        if (okF) {
            struct Job { int bin; size_t at; };
            std::vector<Job> jobs{{0, 0}};
            wF.resize(32, 0.0f);
            const float qnan = std::nanf("");
            while (!jobs.empty()) {
                const Job jb = jobs.back();
                jobs.pop_back();
                for (int h = 0; h < 2; ++h) {
                    const int X = h == 0 ? nd[(size_t)jb.bin].left : nd[(size_t)jb.bin].right;
                    int kids[2] = {X, -1};
                    if (nd[(size_t)X].count <= 0) { kids[0] = nd[(size_t)X].left; kids[1] = nd[(size_t)X].right; }
                    int refs[2] = {RT_NO_CHILD, RT_NO_CHILD};
                    for (int k = 0; k < 2; ++k) {
                        float box[6] = {qnan, qnan, qnan, qnan, qnan, qnan};
                        if (kids[k] >= 0) {
                            const float *b = nodes12 + (size_t)kids[k] * 12;
                            box[0] = b[0]; box[1] = b[1]; box[2] = b[2]; box[3] = b[4]; box[4] = b[5]; box[5] = b[6];
                            if (nd[(size_t)kids[k]].count > 0) refs[k] = refOfW(kids[k]);
                            else {
                                refs[k] = (int)(wF.size() / 32);
                                if ((size_t)refs[k] >= ((size_t)1 << 29)) { okF = false; break; }
                                wF.resize(wF.size() + 32, 0.0f);
                                jobs.push_back({kids[k], (size_t)refs[k]});
                            }
                        }
                        float *o = &wF[jb.at * 32 + (size_t)h * 16 + (size_t)k * 8];
                        o[0] = box[0]; o[1] = box[1]; o[2] = box[2]; o[4] = box[3]; o[5] = box[4]; o[6] = box[5];
                    }
                    if (!okF) break;
                    float *o = &wF[jb.at * 32 + (size_t)h * 16];
                    std::memcpy(&o[3], &refs[0], 4);
                    std::memcpy(&o[7], &refs[1], 4);
                }
                if (!okF) break;
            }
        }
        if (!okF) { wF.clear(); if (wantF && nd[0].count <= 0) c->sceneFlags |= RT_SCENE_NOT_FUSED; }
    }
    // Round 5 -- implicit records.  What the traversal launches cost is their 16-byte lane-loads (one vector-L1 lookup each), and a 64-byte
    // two-child record spends one of its four on two child references.  The reference's builder splits every range at its middle and stops at <= 8 triangles
    // (bvh.cpp:62-76), so whenever n / 2^D falls into [4.5, 8] for some D every leaf sits at depth D and the tree is a perfect binary tree: a node is named by
    // (depth d, path p = the left / right turns from the root as a binary number), its children are (d + 1, 2p) and (d + 1, 2p + 1), the leaves are p = 0 ..
    // 2^D - 1 at depth D -- no reference needs to be stored.  Records: 48 bytes = the two child boxes = THREE loads, at the node's pre-order position
    // d - popcount(p) + (p << (D - d)) (a left child sits next to its parent, as in the reference's own numbering); a leaf's triangle-pair records at
    // p * R (R = the most records any leaf owns), the leaf's triangle count in the spare word of its first record.  Visiting order, boxes and triangle tests
    // are those of the 64-byte records.  Checked here for whatever tree was uploaded (all leaves at one depth <= 23); others keep the explicit records.
    std::vector<float> iN2, iPairs, iN4, iLeafBox;
    std::vector<uint32_t> iQ4;
    int implD = 0, implR = 0;
    if (nd[0].count <= 0 && getenv("RT_IMPLICIT") && atoi(getenv("RT_IMPLICIT")) != 0) {
        struct E { int node; int d; uint32_t p; };
        std::vector<E> st{{0, 0, 0u}};
        std::vector<int> nodeD((size_t)nNodes, -1);
        std::vector<uint32_t> nodeP((size_t)nNodes, 0u);
        int leafDepth = -1, maxRec = 0;
        bool uniform = true;
        while (!st.empty() && uniform) {
            const E e = st.back();
            st.pop_back();
            nodeD[(size_t)e.node] = e.d; nodeP[(size_t)e.node] = e.p;
            const N &n = nd[(size_t)e.node];
            if (n.count > 0) {
                if (leafDepth < 0) leafDepth = e.d;
                if (e.d != leafDepth) uniform = false;
                maxRec = std::max(maxRec, (n.count + 1) / 2);
            } else {
                if (e.d >= 23) { uniform = false; break; }
                st.push_back({n.left, e.d + 1, e.p * 2u});
                st.push_back({n.right, e.d + 1, e.p * 2u + 1u});
            }
        }
        if (uniform && leafDepth >= 1) {
            const int D = leafDepth;
            iN2.assign((((size_t)1 << D) - 1) * 12, 0.0f);
            iPairs.assign((((size_t)1 << D) * (size_t)maxRec + 8) * 20, 0.0f);
            for (int i = 0; i < nNodes; ++i) {
                if (nodeD[(size_t)i] < 0) continue;             // (unreachable nodes: none in a valid tree)
                const N &n = nd[(size_t)i];
                const int d = nodeD[(size_t)i];
                const uint32_t pth = nodeP[(size_t)i];
                if (n.count <= 0) {
                    const size_t at = (size_t)d - (size_t)__builtin_popcount(pth) + ((size_t)pth << (D - d));
                    const float *L = nodes12 + (size_t)n.left * 12, *R = nodes12 + (size_t)n.right * 12;
                    const float rec[12] = {L[0], L[1], L[2], L[4], L[5], L[6], R[0], R[1], R[2], R[4], R[5], R[6]};
                    std::memcpy(&iN2[at * 12], rec, sizeof rec);
                } else {
                    // the leaf's records as `pairs` holds them, at p * R; the count in the spare word of the first
                    const size_t src = (size_t)(-pairRefOf[(size_t)i] - 1) >> 3, nrec = (size_t)(n.count + 1) / 2;
                    float *o = &iPairs[(size_t)pth * (size_t)maxRec * 20];
                    std::memcpy(o, &pairs[src * 20], nrec * 20 * sizeof(float));
                    const uint32_t cnt = (uint32_t)n.count;
                    std::memcpy(&o[19], &cnt, 4);
                }
            }
            implD = D; implR = maxRec;
            // ... and the any-hit walk's four-wide records in the same naming: an even-depth node (d, p) has the children (d + 2, 4p + j), j = 0..3 (or, when d + 1 == D,
            // the two leaves 2p, 2p + 1), so the record is the four child boxes alone, component-wise: 96 bytes, SIX loads instead of seven, at the node's own
            // pre-order position (odd-depth slots of the array stay empty and are never touched).  Where the quantised form is walked (trees beyond the L2) the same
            // record in bytes: [origin.xyz, exponents] [lo.x lo.y lo.z hi.x] [hi.y hi.z - -] = 48 bytes, THREE loads instead of four, and the leaves' exact boxes at
            // their ordinal p.
            const bool wantQ = !q4.empty();
            const float qnan = std::nanf("");
            iN4.assign((((size_t)1 << D) - 1) * 24, 0.0f);
            if (wantQ) { iQ4.assign((((size_t)1 << D) - 1) * 12, 0u); iLeafBox.assign(((size_t)1 << D) * 8, 0.0f); }
            bool okI = true;
            for (int i = 0; i < nNodes && okI; ++i) {
                const int d = nodeD[(size_t)i];
                if (d < 0) continue;
                const N &n = nd[(size_t)i];
                const uint32_t pth = nodeP[(size_t)i];
                if (n.count > 0) {
                    if (wantQ) { const float *b = nodes12 + (size_t)i * 12; const float box[8] = {b[0], b[1], b[2], b[4], b[5], b[6], 0.0f, 0.0f}; std::memcpy(&iLeafBox[(size_t)pth * 8], box, sizeof box); }
                    continue;
                }
                if (d & 1) continue;
                int kids[4] = {-1, -1, -1, -1};
                if (d + 1 == D) { kids[0] = n.left; kids[1] = n.right; }
                else { kids[0] = nd[(size_t)n.left].left; kids[1] = nd[(size_t)n.left].right; kids[2] = nd[(size_t)n.right].left; kids[3] = nd[(size_t)n.right].right; }
                const size_t at = (size_t)d - (size_t)__builtin_popcount(pth) + ((size_t)pth << (D - d));
                float *o = &iN4[at * 24];
                for (int k = 0; k < 4; ++k) {
                    const float *b = kids[k] >= 0 ? nodes12 + (size_t)kids[k] * 12 : nullptr;
                    o[0 + k] = b ? b[0] : qnan; o[4 + k] = b ? b[1] : qnan; o[8 + k] = b ? b[2] : qnan;
                    o[12 + k] = b ? b[4] : qnan; o[16 + k] = b ? b[5] : qnan; o[20 + k] = b ? b[6] : qnan;
                }
                if (!wantQ) continue;
                // quantise as the explicit form does (above): origin = the children's common minimum, one power-of-two step per axis, bytes moved outward until the decoded
                // box -- in the kernel's own expression fmaf(byte, 2^e, origin) -- contains the child's
                uint32_t *q = &iQ4[at * 12];
                float org[3], scale[3];
                uint32_t exps = 0;
                for (int a = 0; a < 3 && okI; ++a) {
                    float lo = INFINITY, hi = -INFINITY;
                    for (int k = 0; k < 4; ++k) if (kids[k] >= 0) { lo = std::min(lo, o[4 * a + k]); hi = std::max(hi, o[12 + 4 * a + k]); }
                    if (!(lo <= hi)) { lo = hi = 0.0f; }
                    int eb = 1;
                    const double ext = ((double)hi - (double)lo) / 255.0;
                    if (ext > 0.0) { int e2; (void)std::frexp(ext, &e2); eb = std::max(1, e2 - 1 + 127); }
                    while (eb <= 254 && std::fmaf(255.0f, std::ldexp(1.0f, eb - 127), lo) < hi) ++eb;
                    if (eb > 254) { okI = false; break; }
                    org[a] = lo; scale[a] = std::ldexp(1.0f, eb - 127);
                    exps |= (uint32_t)eb << (8 * a);
                    std::memcpy(&q[a], &lo, 4);
                }
                if (!okI) break;
                q[3] = exps;
                for (int k = 0; k < 4; ++k) {
                    if (kids[k] < 0) continue;
                    for (int a = 0; a < 3; ++a) {
                        const float lo = o[4 * a + k], hi = o[12 + 4 * a + k];
                        int ql = (int)std::floor(((double)lo - (double)org[a]) / (double)scale[a]);
                        ql = std::max(0, std::min(255, ql));
                        while (ql > 0 && std::fmaf((float)ql, scale[a], org[a]) > lo) --ql;
                        int qh = (int)std::ceil(((double)hi - (double)org[a]) / (double)scale[a]);
                        qh = std::max(0, std::min(255, qh));
                        while (qh < 255 && std::fmaf((float)qh, scale[a], org[a]) < hi) ++qh;
                        if (std::fmaf((float)ql, scale[a], org[a]) > lo || std::fmaf((float)qh, scale[a], org[a]) < hi) okI = false;
                        const int wl = 4 + a, wh = a == 0 ? 7 : 7 + a;      // words: lo.x lo.y lo.z hi.x | hi.y hi.z
                        q[wl] |= (uint32_t)ql << (8 * k);
                        q[wh] |= (uint32_t)qh << (8 * k);
                    }
                }
            }
            if (!okI) { iQ4.clear(); iLeafBox.clear(); }      // (the explicit quantised form passed the same checks, so this does not happen)
        }
    }
    if (depth > 32) return fail(c, RT_ERR_UNSUPPORTED, "rt_upload_bvh: tree depth %d exceeds the 32-entry traversal stack", depth);
    if (!q4.empty()) {
        HIP_TRY(c, hipMalloc(&c->dQ4, q4.size() * 4));
        HIP_TRY(c, hipMemcpy(c->dQ4, q4.data(), q4.size() * 4, hipMemcpyHostToDevice));
        HIP_TRY(c, hipMalloc(&c->dLeafBox, leafBox.size() * 4));
        HIP_TRY(c, hipMemcpy(c->dLeafBox, leafBox.data(), leafBox.size() * 4, hipMemcpyHostToDevice));
    }
    if (!iN2.empty()) {
        HIP_TRY(c, hipMalloc(&c->dIN2, iN2.size() * sizeof(float)));
        HIP_TRY(c, hipMemcpy(c->dIN2, iN2.data(), iN2.size() * sizeof(float), hipMemcpyHostToDevice));
        HIP_TRY(c, hipMalloc(&c->dIPairs, iPairs.size() * sizeof(float)));
        HIP_TRY(c, hipMemcpy(c->dIPairs, iPairs.data(), iPairs.size() * sizeof(float), hipMemcpyHostToDevice));
        c->implD = implD; c->implR = implR;
        HIP_TRY(c, hipMalloc(&c->dIN4, iN4.size() * sizeof(float)));
        HIP_TRY(c, hipMemcpy(c->dIN4, iN4.data(), iN4.size() * sizeof(float), hipMemcpyHostToDevice));
        if (!iQ4.empty()) {
            HIP_TRY(c, hipMalloc(&c->dIQ4, iQ4.size() * 4));
            HIP_TRY(c, hipMemcpy(c->dIQ4, iQ4.data(), iQ4.size() * 4, hipMemcpyHostToDevice));
            HIP_TRY(c, hipMalloc(&c->dILeafBox, iLeafBox.size() * 4));
            HIP_TRY(c, hipMemcpy(c->dILeafBox, iLeafBox.data(), iLeafBox.size() * 4, hipMemcpyHostToDevice));
        }
    }
    if (!wF.empty()) {
        HIP_TRY(c, hipMalloc(&c->dWF, wF.size() * sizeof(float)));
        HIP_TRY(c, hipMemcpy(c->dWF, wF.data(), wF.size() * sizeof(float), hipMemcpyHostToDevice));
        c->nFused = wF.size() / 32;
    }
    HIP_TRY(c, hipMalloc(&c->dWNodes, wn.size() * sizeof(float)));
    HIP_TRY(c, hipMalloc(&c->dW4, w4.size() * sizeof(float)));
    HIP_TRY(c, hipMemcpy(c->dW4, w4.data(), w4.size() * sizeof(float), hipMemcpyHostToDevice));
    c->rootRef4 = rootRef4;
    c->anyStack = anyStack;
    c->rootRefW = refOfW(0);
    HIP_TRY(c, hipMalloc(&c->dWNodesW, wnW.size() * sizeof(float)));
    HIP_TRY(c, hipMemcpy(c->dWNodesW, wnW.data(), wnW.size() * sizeof(float), hipMemcpyHostToDevice));
    HIP_TRY(c, hipMalloc(&c->dPairs, pairs.size() * sizeof(float)));
    HIP_TRY(c, hipMemcpy(c->dPairs, pairs.data(), pairs.size() * sizeof(float), hipMemcpyHostToDevice));
    // 8 triangles of zero padding: the traversal kernels load triangle records in groups without a bounds branch
    HIP_TRY(c, hipMalloc(&c->dTris, (size_t)(nTris + 8) * 12 * sizeof(float)));
    HIP_TRY(c, hipMemset(c->dTris, 0, (size_t)(nTris + 8) * 12 * sizeof(float)));
    HIP_TRY(c, hipMemcpy(c->dWNodes, wn.data(), wn.size() * sizeof(float), hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(c->dTris, tris12, (size_t)nTris * 12 * sizeof(float), hipMemcpyHostToDevice));
    c->nNodes = nNodes; c->nTris = nTris; c->nInner = nInner; c->treeDepth = depth;
    c->nWide4 = w4.size() / 32; c->nPairs = pairs.size() / 20 - 8;
    c->rootRef = refOf(0);
    std::memcpy(c->rootMin, nodes12, 12);
    std::memcpy(c->rootMax, nodes12 + 4, 12);
    return RT_OK;
    });
}

int rt_build_bvh_gpu(RtContext *c, const float *tris9, int nTris, float *nodes12, float *tris12) {
    if (!c) return RT_ERR_INVALID;
    const char *err = nullptr;
    const int rc = rtl::build_bvh_gpu(c->cfg.device, tris9, nTris, nodes12, tris12, &err);
    if (rc < 0) return fail(c, rc, "rt_build_bvh_gpu: %s", err ? err : "bad arguments");
    return rc;
}

int rt_upload_env(RtContext *c, const uint8_t *faces, int faceSize, int channels) {
    if (!c) return RT_ERR_INVALID;
    (void)hipSetDevice(c->cfg.device);
    static const uint8_t dummy[6 * 4] = {128, 128, 255, 255, 128, 128, 255, 255, 128, 128, 255, 255,
                                         128, 128, 255, 255, 128, 128, 255, 255, 128, 128, 255, 255};   // cubemap.cpp:13
    if (!faces) { faces = dummy; faceSize = 1; channels = 4; }
    if (faceSize <= 0 || (channels != 3 && channels != 4)) return fail(c, RT_ERR_INVALID, "rt_upload_env: faceSize=%d channels=%d", faceSize, channels);
    return guarded(c, "rt_upload_env", [&]() -> int {
    const size_t texels = (size_t)6 * faceSize * faceSize;
    std::vector<uint8_t> rgba(texels * 4);
    for (size_t i = 0; i < texels; ++i) {
        rgba[i * 4 + 0] = faces[i * channels + 0];
        rgba[i * 4 + 1] = faces[i * channels + 1];
        rgba[i * 4 + 2] = faces[i * channels + 2];
        rgba[i * 4 + 3] = (channels == 4) ? faces[i * channels + 3] : (uint8_t)255;
    }
    HIP_TRY(c, sync_all(c));
    if (c->dEnv) (void)hipFree(c->dEnv);
    c->dEnv = nullptr;
    HIP_TRY(c, hipMalloc(&c->dEnv, texels * 4));
    HIP_TRY(c, hipMemcpy(c->dEnv, rgba.data(), texels * 4, hipMemcpyHostToDevice));
    c->envSize = faceSize;
    return RT_OK;
    });
}

int rt_resize(RtContext *c, int w, int h) {
    if (!c) return RT_ERR_INVALID;
    if (w <= 0 || h <= 0) return fail(c, RT_ERR_INVALID, "rt_resize: %dx%d", w, h);
    (void)hipSetDevice(c->cfg.device);
    HIP_TRY(c, sync_all(c));
    free_targets(c);
    FrameGeom g;
    g.W = w; g.H = h;
    g.tilesX = (w + RT_TILE_DIM - 1) / RT_TILE_DIM;
    g.tilesY = (h + RT_TILE_DIM - 1) / RT_TILE_DIM;
    g.nTiles = g.tilesX * g.tilesY;
    g.rank = c->cfg.rank; g.world = c->cfg.worldSize;
    g.nLocalTiles = (g.nTiles - g.rank + g.world - 1) / g.world;
    if (g.nLocalTiles < 0) g.nLocalTiles = 0;
    g.batch = 1;
    c->g = g;
    // every rank allocates the padded size so gather blocks are equal
    const size_t maxLocal = (size_t)(g.nTiles + g.world - 1) / g.world;
    c->nSlots = std::max<size_t>(maxLocal, 1) * RT_TILE_PIXELS;
    for (int i = 0; i < c->nLanes; ++i) HIP_TRY(c, hipMalloc(&c->dColor[i], c->nSlots * 8));
    for (int i = 0; i < c->nLanes; ++i) {
        HIP_TRY(c, hipMalloc(&c->dMotion[i], c->nSlots * 4));
        HIP_TRY(c, hipMalloc(&c->dGPos[i], c->nSlots * 8));
        HIP_TRY(c, hipMalloc(&c->dGNrm[i], c->nSlots * 8));
    }
    c->sized = true;
    c->haveFrameState = false;
    return rt_reset_accum(c);
}

int rt_reset_accum(RtContext *c) {
    if (!c) return RT_ERR_INVALID;
    if (!c->sized) return fail(c, RT_ERR_STATE, "rt_reset_accum before rt_resize");
    (void)hipSetDevice(c->cfg.device);
    HIP_TRY(c, sync_all(c));
    c->frameIndex = 0;
    c->writeIdx = 0;
    for (int i = 0; i < RT_MAX_LANES; ++i) c->histExchanged[i] = false;
    for (int i = 0; i < c->nLanes; ++i) HIP_TRY(c, hipMemsetAsync(c->dColor[i], 0, c->nSlots * 8, c->stream));
    for (int i = 0; i < c->nLanes; ++i) {
        HIP_TRY(c, hipMemsetAsync(c->dMotion[i], 0, c->nSlots * 4, c->stream));
        HIP_TRY(c, hipMemsetAsync(c->dGPos[i], 0, c->nSlots * 8, c->stream));
        HIP_TRY(c, hipMemsetAsync(c->dGNrm[i], 0, c->nSlots * 8, c->stream));
    }
    for (int w = 0; w < 4; ++w) c->gatheredLane[w] = -1;
    return RT_OK;
}

int rt_frame_index(const RtContext *c) { return c ? c->frameIndex : RT_ERR_INVALID; }

// One set of launches for `batch` consecutive frames (batch == 1: the plain frame).  jitterK: uJitter of the batch's frames.
static int render_frames_impl(RtContext *c, const RtUniforms *uIn, int batch, const float (*jitterK)[2]) {
    if (!c || !uIn) return RT_ERR_INVALID;
    if (!c->sized) return fail(c, RT_ERR_STATE, "rt_render_frame before rt_resize");
    (void)hipSetDevice(c->cfg.device);
    if (c->timing) harvest_stage_events(c);
    DevFrame fr;
    fr.u = *uIn;
    fr.u.frameIndex = c->frameIndex;
    if ((int)fr.u.resolution[0] != c->g.W || (int)fr.u.resolution[1] != c->g.H)
        return fail(c, RT_ERR_INVALID, "rt_render_frame: uResolution %gx%g != framebuffer %dx%d", fr.u.resolution[0], fr.u.resolution[1], c->g.W, c->g.H);
    if ((fr.u.useBVH == 1 || fr.u.useBVH == RT_SCENE_HYBRID) && fr.u.nodeCount > 0 && fr.u.triCount > 0 && (c->nNodes == 0 || fr.u.nodeCount > c->nNodes || fr.u.triCount > c->nTris))
        return fail(c, RT_ERR_STATE, "rt_render_frame: uniforms name %d nodes / %d tris, uploaded %d / %d", fr.u.nodeCount, fr.u.triCount, c->nNodes, c->nTris);
    if (fr.u.useEnvMap == 1 && !c->dEnv) return fail(c, RT_ERR_STATE, "rt_render_frame: uUseEnvMap without an environment");
    const int prevLaneX = (c->writeIdx + c->nLanes - 1) % c->nLanes;
    const bool needAll = fr.u.cameraMoved == 1 && c->g.world > 1 && fr.u.enableTAA == 1 && c->frameIndex > 0;
    if (needAll && !(c->dHistAll[prevLaneX] && c->histExchanged[prevLaneX]))
        return fail(c, RT_ERR_STATE, "rt_render_frame: cameraMoved on a tile-parallel context: reprojection reads other ranks' history -- all-gather the "
                                     "previous frame's COLOR0 blocks into rt_history_exchange_buffer() and call rt_history_exchanged() first");
    fr.sc = make_dev_scene(c);
    if (!(fr.u.nodeCount > 0 && fr.u.triCount > 0)) fr.sc.hasBVH = 0;
    fr.g = c->g;
    fr.g.batch = batch;
    for (int k = 0; k < RT_MAX_BATCH; ++k) { fr.jitterK[k][0] = jitterK ? jitterK[k < batch ? k : 0][0] : fr.u.jitter[0]; fr.jitterK[k][1] = jitterK ? jitterK[k < batch ? k : 0][1] : fr.u.jitter[1]; }
    fr.giBounces = c->giBounces;
    // Lane = frame index mod nLanes = index of the COLOR0 buffer this frame writes: consecutive frames rotate over the lanes'
    // streams and overlap everywhere except at the temporal resolve (the only read of the previous frame), and every later
    // reader of a COLOR0 buffer (gather, assemble) is stream-ordered before the next writer of the same buffer.
    const int lane = c->writeIdx, prevLane = (c->writeIdx + c->nLanes - 1) % c->nLanes;
    hipStream_t st = c->lanes[lane];
    if (c->serialFrames) HIP_TRY(c, hipStreamWaitEvent(st, c->evDone[prevLane], 0));
    HIP_TRY(c, hipMemcpyAsync(c->dFrame[lane], &fr, sizeof(fr), hipMemcpyHostToDevice, st));
    Targets tg;
    tg.color = c->dColor[c->writeIdx];
    tg.prev = c->dColor[prevLane];
    tg.prevAll = needAll ? (const uint2 *)c->dHistAll[prevLane] : nullptr;
    tg.blockSlots = (int)c->nSlots;
    c->histExchanged[lane] = false;   // this lane's exchange buffer belongs to the frame that is about to be rendered
    tg.motion = c->dMotion[lane]; tg.gpos = c->dGPos[lane]; tg.gnrm = c->dGNrm[lane];
    const bool count = c->cfg.countWork != 0;
    int pipeline = c->cfg.pipeline;
    if (pipeline == RT_PIPELINE_AUTO) pipeline = (fr.u.useBVH == 1 && fr.sc.hasBVH && !count) ? RT_PIPELINE_WAVEFRONT : RT_PIPELINE_MEGAKERNEL;
    if (pipeline == RT_PIPELINE_WAVEFRONT && !(fr.u.useBVH == 1)) pipeline = RT_PIPELINE_MEGAKERNEL;   // analytic scene: pure ALU, megakernel only
    if (batch > 1 && pipeline != RT_PIPELINE_WAVEFRONT) return fail(c, RT_ERR_STATE, "internal: a frame batch reached the megakernel");
    // EXTENSION: the hybrid scene in stages (rt_hybrid.hip) unless the megakernel was asked for; same frames bit for bit
    const bool staged = c->cfg.pipeline != RT_PIPELINE_MEGAKERNEL && fr.u.useBVH == RT_SCENE_HYBRID && fr.sc.hasBVH && !count;
    if (staged) {
        if (!c->hybrid[0]) c->hybrid[0] = rt_hybrid_create(c->cus);   // one arena for all lanes: the passes synchronise with the host anyway
        int rc = rt_hybrid_render(c->hybrid[0], c, st, c->dFrame[lane], fr, tg, std::max(c->treeDepth, 1), c->nLanes > 1 ? c->evDone[prevLane] : nullptr);
        if (rc != RT_OK) return fail(c, rc, "staged hybrid pipeline: %s", rt_hybrid_error(c->hybrid[0]));
    } else if (pipeline == RT_PIPELINE_WAVEFRONT) {
        int rc = rt_wave_render(c->wave[lane], c, st, c->dFrame[lane], fr, tg, c->dCounters, count, std::max(c->treeDepth, 1), c->nLanes > 1 ? c->evDone[prevLane] : nullptr,
                                (size_t)c->nInner * 64 + c->nWide4 * 128 + c->nPairs * 80 < ((size_t)32 << 20));
        if (rc != RT_OK) return fail(c, rc, "wavefront pipeline: %s", rt_wave_error(c->wave[lane]));
    } else {
        if (c->nLanes > 1) HIP_TRY(c, hipStreamWaitEvent(st, c->evDone[prevLane], 0));   // the megakernel reads the history from its first instruction on
        rt_stage_begin(c, 0, st);
        HIP_TRY(c, rtl::launch_mega(st, c->dFrame[lane], tg, c->dCounters, count, std::max(c->treeDepth, 1), c->g.nLocalTiles));
        rt_stage_end(c, 0, 1, st);
    }
    HIP_TRY(c, hipEventRecord(c->evDone[lane], st));
    c->lastStream = st;
    if (c->timing) c->timedFrames += batch;
    c->frameIndex += batch;          // Accum::swapAfterFrame, include/render/accum.h:125-128
    c->writeIdx = (c->writeIdx + 1) % c->nLanes;
    return RT_OK;
}

int rt_render_frame(RtContext *c, const RtUniforms *uIn) { return render_frames_impl(c, uIn, 1, nullptr); }

// K consecutive frames of a static camera in ONE set of launches ("batch several frames per gather ... when the camera is
// static").  The frames of such a sequence differ only in uFrameIndex and uJitter; the only thing frame f+1 needs from frame f is the
// accumulation history at its own pixel (rt_taa.glsl:86-105), which the resolve kernel chains in registers.  What a tile-parallel
// rank gains: each launch carries K times the work, so the fixed cost of the persistent traversal launches (ramp-up, tail of the
// longest rays) and of nine launches per frame is paid once per K frames -- measured per rank at world = 8: 0.40 -> 0.27 ms per frame
// at K = 4.  Results are bit-identical to K calls of rt_render_frame; the four targets afterwards are those of the LAST frame.
int rt_render_frames(RtContext *c, const RtUniforms *us, int count) {
    if (!c || !us || count < 1) return RT_ERR_INVALID;
    if (!c->sized) return fail(c, RT_ERR_STATE, "rt_render_frames before rt_resize");
    // A run of frames shares one set of launches only if every frame of it would take the wavefront pipeline (decided per run from
    // its first frame: the uniform blocks of a run agree in everything but jitter, so they agree in useBVH / nodeCount / triCount) and
    // if the resolve of frame k > 0 is sure to take the still branch of resolveTAA (rt_taa.glsl:86-105), whose history is the pixel's
    // own texel and is chained in registers.  With uTaaStillThresh <= 0 the test `length(motion) < thresh` fails even for zero motion
    // and the reprojection branch would read the texture from BEFORE the batch: such frames go one by one.
    auto batchable = [&](const RtUniforms &u) {
        return c->cfg.pipeline != RT_PIPELINE_MEGAKERNEL && c->cfg.countWork == 0 && u.useBVH == 1 && c->nNodes > 0 && u.nodeCount > 0 && u.triCount > 0 &&
               u.cameraMoved == 0 && (u.enableTAA == 0 || u.taaStillThresh > 0.0f);
    };
    // uniform blocks of a batch must agree in everything but frameIndex (ignored anyway) and jitter
    auto same_but_jitter = [](const RtUniforms &a, const RtUniforms &b) {
        RtUniforms x = a, y = b;
        x.frameIndex = y.frameIndex = 0;
        x.jitter[0] = y.jitter[0] = x.jitter[1] = y.jitter[1] = 0.0f;
        return std::memcmp(&x, &y, sizeof x) == 0;
    };
    int done = 0;
    while (done < count) {
        int k = 1;
        if (batchable(us[done]))
            while (k < RT_MAX_BATCH && done + k < count && same_but_jitter(us[done], us[done + k])) ++k;
        float jit[RT_MAX_BATCH][2] = {};
        for (int q = 0; q < k; ++q) { jit[q][0] = us[done + q].jitter[0]; jit[q][1] = us[done + q].jitter[1]; }
        int rc = render_frames_impl(c, &us[done], k, jit);
        if (rc != RT_OK) return rc;
        done += k;
    }
    return RT_OK;
}

int rt_render_ray(RtContext *c, const RtRenderParams *params, const RtCamera *cam, int useBVH, int showMotion, const float *currView,
                  const float *currProj) {
    if (!c || !params || !cam) return RT_ERR_INVALID;
    if (!c->sized) return fail(c, RT_ERR_STATE, "rt_render_ray before rt_resize");
    float V[16], P[16], VP[16];
    if (currView) std::memcpy(V, currView, 64); else rt_camera_view(cam, V);
    if (currProj) std::memcpy(P, currProj, 64); else rt_camera_proj(cam, P);
    rt_mat4_mul(P, V, VP);                                   // FrameState::beginFrame, frame_state.h:68-73
    if (!c->haveFrameState) { std::memcpy(c->prevVP, VP, 64); c->haveFrameState = true; }   // application.cpp:316-319
    const int moved = rt_camera_moved(VP, c->prevVP);        // application.cpp:387-395
    RtUniforms u;
    rt_make_uniforms(params, cam, V, VP, c->prevVP, c->g.W, c->g.H, c->frameIndex, moved, useBVH, showMotion, c->nNodes, c->nTris,
                     c->dEnv != nullptr, &u);
    int rc = rt_render_frame(c, &u);
    if (rc != RT_OK) return rc;
    std::memcpy(c->prevVP, VP, 64);                          // FrameState::endFrame, frame_state.h:81-84
    return RT_OK;
}

int rt_set_extension(RtContext *c, const RtExtension *ext) {
    if (!c || !ext) return RT_ERR_INVALID;
    if (ext->giBounces < 1 || ext->giBounces > 8) return fail(c, RT_ERR_INVALID, "rt_set_extension: giBounces = %d (1..8)", ext->giBounces);
    if (ext->envFilter != 0 && ext->envFilter != 1) return fail(c, RT_ERR_INVALID, "rt_set_extension: envFilter = %d (0 or 1)", ext->envFilter);
    c->giBounces = ext->giBounces;
    c->envFilter = ext->envFilter;
    return RT_OK;
}

// `count` consecutive rt_render_ray calls with the same camera and parameters, handed to rt_render_frames as one sequence (so an
// accumulating static camera is rendered in batches).  Frame state (prev / curr view-projection) is kept exactly as rt_render_ray does.
int rt_render_ray_frames(RtContext *c, const RtRenderParams *params, const RtCamera *cam, int useBVH, int showMotion, int count) {
    if (!c || !params || !cam || count < 1) return RT_ERR_INVALID;
    if (!c->sized) return fail(c, RT_ERR_STATE, "rt_render_ray_frames before rt_resize");
    return guarded(c, "rt_render_ray_frames", [&]() -> int {
        float V[16], P[16], VP[16];
        rt_camera_view(cam, V);
        rt_camera_proj(cam, P);
        rt_mat4_mul(P, V, VP);
        if (!c->haveFrameState) { std::memcpy(c->prevVP, VP, 64); c->haveFrameState = true; }
        std::vector<RtUniforms> us((size_t)count);
        float prev[16];
        std::memcpy(prev, c->prevVP, 64);
        for (int i = 0; i < count; ++i) {
            const int moved = rt_camera_moved(VP, prev);
            rt_make_uniforms(params, cam, V, VP, prev, c->g.W, c->g.H, c->frameIndex + i, moved, useBVH, showMotion, c->nNodes, c->nTris, c->dEnv != nullptr,
                             &us[(size_t)i]);
            std::memcpy(prev, VP, 64);
        }
        const int first = c->frameIndex;
        int rc = rt_render_frames(c, us.data(), count);
        // FrameState::endFrame (frame_state.h:81-84) runs after every rendered frame: if the sequence failed part-way, the frames
        // that did render have advanced frameIndex and the camera state must follow them, exactly as with rt_render_ray per frame
        if (rc == RT_OK || c->frameIndex != first) std::memcpy(c->prevVP, VP, 64);
        return rc;
    });
}

int rt_synchronize(RtContext *c) {
    if (!c) return RT_ERR_INVALID;
    (void)hipSetDevice(c->cfg.device);
    HIP_TRY(c, sync_all(c));
    return RT_OK;
}

int rt_read_target(RtContext *c, int which, void *dst, int fmt) {
    if (!c || !dst) return RT_ERR_INVALID;
    if (!c->sized) return fail(c, RT_ERR_STATE, "rt_read_target before rt_resize");
    (void)hipSetDevice(c->cfg.device);
    int ch;
    void *src = target_ptr(c, which, ch);
    if (!src || (fmt != RT_FORMAT_F16 && fmt != RT_FORMAT_F32)) return fail(c, RT_ERR_INVALID, "rt_read_target: which=%d format=%d", which, fmt);
    const size_t n = (size_t)c->g.W * c->g.H, bytes = n * ch * (fmt == RT_FORMAT_F32 ? 4 : 2);
    HIP_TRY(c, sync_all(c));
    int rc = ensure_staging(c, bytes);
    if (rc != RT_OK) return rc;
    hipLaunchKernelGGL(k_untile, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, src, c->dStaging, c->g, ch, fmt == RT_FORMAT_F32 ? 1 : 0);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(dst, c->dStaging, bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, sync_all(c));
    return RT_OK;
}

int rt_write_target(RtContext *c, int which, const void *srcHost, int fmt) {
    if (!c || !srcHost) return RT_ERR_INVALID;
    if (!c->sized) return fail(c, RT_ERR_STATE, "rt_write_target before rt_resize");
    (void)hipSetDevice(c->cfg.device);
    int ch;
    void *dstT = target_ptr(c, which, ch);
    if (!dstT || fmt != RT_FORMAT_F16) return fail(c, RT_ERR_INVALID, "rt_write_target: which=%d format=%d (RT_FORMAT_F16 only)", which, fmt);
    const size_t bytes = (size_t)c->g.W * c->g.H * ch * 2;
    HIP_TRY(c, sync_all(c));
    int rc = ensure_staging(c, bytes);
    if (rc != RT_OK) return rc;
    HIP_TRY(c, hipMemcpyAsync(c->dStaging, srcHost, bytes, hipMemcpyHostToDevice, c->stream));
    const unsigned nSlots = (unsigned)c->g.nLocalTiles * 256u;
    hipLaunchKernelGGL(k_tile, dim3((nSlots + 255) / 256), dim3(256), 0, c->stream, (const void *)c->dStaging, dstT, c->g, ch);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, sync_all(c));
    return RT_OK;
}

int rt_present(RtContext *c, const RtPresentParams *p, uint8_t *dst) {
    if (!c || !p || !dst) return RT_ERR_INVALID;
    if (!c->sized) return fail(c, RT_ERR_STATE, "rt_present before rt_resize");
    if (c->g.world > 1) return fail(c, RT_ERR_UNSUPPORTED, "rt_present: the 7x7 filter reads other ranks' tiles; gather the four targets and use rt_present_gathered");
    if ((int)p->resolution[0] != c->g.W || (int)p->resolution[1] != c->g.H) return fail(c, RT_ERR_INVALID, "rt_present: uResolution != framebuffer");
    (void)hipSetDevice(c->cfg.device);
    const size_t bytes = (size_t)c->g.W * c->g.H * 4;
    HIP_TRY(c, sync_all(c));
    int rc = ensure_staging(c, bytes);
    if (rc != RT_OK) return rc;
    rt_stage_begin(c, 11);
    HIP_TRY(c, rtl::launch_present(c->stream, c->g, c->dColor[last_lane(c)], c->dMotion[last_lane(c)], c->dGPos[last_lane(c)], c->dGNrm[last_lane(c)], *p, (uint32_t *)c->dStaging));
    rt_stage_end(c, 11, 1);
    HIP_TRY(c, hipMemcpyAsync(dst, c->dStaging, bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, sync_all(c));
    return RT_OK;
}

int rt_present_gathered(RtContext *c, const RtPresentParams *p, const void *color, const void *motion, const void *gpos, const void *gnrm, uint8_t *dst) {
    if (!c || !p || !color || !motion || !gpos || !gnrm || !dst) return RT_ERR_INVALID;
    if (!c->sized) return fail(c, RT_ERR_STATE, "rt_present_gathered before rt_resize");
    if ((int)p->resolution[0] != c->g.W || (int)p->resolution[1] != c->g.H) return fail(c, RT_ERR_INVALID, "rt_present_gathered: uResolution != framebuffer");
    (void)hipSetDevice(c->cfg.device);
    const size_t bytes = (size_t)c->g.W * c->g.H * 4;
    HIP_TRY(c, sync_all(c));
    int rc = ensure_staging(c, bytes);
    if (rc != RT_OK) return rc;
    rt_stage_begin(c, 11);
    HIP_TRY(c, rtl::launch_present(c->stream, c->g, (const uint2 *)color, (const uint32_t *)motion, (const uint2 *)gpos, (const uint2 *)gnrm, *p,
                                   (uint32_t *)c->dStaging, (int)c->nSlots));
    rt_stage_end(c, 11, 1);
    HIP_TRY(c, hipMemcpyAsync(dst, c->dStaging, bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, sync_all(c));
    return RT_OK;
}

int rt_history_exchange_buffer(RtContext *c, void **devPtr, size_t *bytes) {
    if (!c || !devPtr || !bytes) return RT_ERR_INVALID;
    if (!c->sized) return fail(c, RT_ERR_STATE, "rt_history_exchange_buffer before rt_resize");
    (void)hipSetDevice(c->cfg.device);
    const int lane = (c->writeIdx + c->nLanes - 1) % c->nLanes;   // the frame rendered last
    const size_t n = (size_t)c->g.world * c->nSlots * 8;
    if (!c->dHistAll[lane]) HIP_TRY(c, hipMalloc(&c->dHistAll[lane], n));
    *devPtr = c->dHistAll[lane];
    *bytes = n;
    return RT_OK;
}
int rt_history_exchanged(RtContext *c) {
    if (!c) return RT_ERR_INVALID;
    if (!c->sized) return fail(c, RT_ERR_STATE, "rt_history_exchanged before rt_resize");
    (void)hipSetDevice(c->cfg.device);
    const int lane = (c->writeIdx + c->nLanes - 1) % c->nLanes;
    if (!c->dHistAll[lane]) return fail(c, RT_ERR_STATE, "rt_history_exchanged without rt_history_exchange_buffer");
    // the next frame's resolve waits on this event: it now also covers the all-gather the caller enqueued on rt_stream()
    HIP_TRY(c, hipEventRecord(c->evDone[lane], c->lanes[lane]));
    c->histExchanged[lane] = true;
    return RT_OK;
}

int rt_local_target(RtContext *c, int which, void **devPtr, size_t *bytes) {
    if (!c || !devPtr || !bytes) return RT_ERR_INVALID;
    if (!c->sized) return fail(c, RT_ERR_STATE, "rt_local_target before rt_resize");
    int ch;
    void *p = target_ptr(c, which, ch);
    if (!p) return fail(c, RT_ERR_INVALID, "rt_local_target: which=%d", which);
    *devPtr = p;
    *bytes = c->nSlots * ch * 2;
    return RT_OK;
}
int rt_gather_block_bytes(const RtContext *c, int which, size_t *bytes) {
    if (!c || !bytes || !c->sized) return RT_ERR_INVALID;
    const int ch = (which == RT_TARGET_MOTION) ? 2 : 4;
    *bytes = c->nSlots * ch * 2;
    return RT_OK;
}
int rt_assemble_gathered(RtContext *c, int which, const void *gatheredDev, void *dstDev) {
    if (!c || !gatheredDev || !dstDev) return RT_ERR_INVALID;
    if (!c->sized) return fail(c, RT_ERR_STATE, "rt_assemble_gathered before rt_resize");
    (void)hipSetDevice(c->cfg.device);
    const int ch = (which == RT_TARGET_MOTION) ? 2 : 4;
    const size_t n = (size_t)c->g.W * c->g.H;
    hipStream_t st = c->lastStream ? c->lastStream : c->stream;   // behind the gather the caller enqueued on rt_stream()
    rt_stage_begin(c, 10, st);
    hipLaunchKernelGGL(k_assemble, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, gatheredDev, dstDev, c->g, ch, c->nSlots * ch * 2);
    rt_stage_end(c, 10, 1, st);
    HIP_TRY(c, hipGetLastError());
    return RT_OK;
}
int rt_stream(RtContext *c, void **s) {
    if (!c || !s) return RT_ERR_INVALID;
    *s = (void *)(c->lastStream ? c->lastStream : c->stream);   // the stream the most recent frame was enqueued on
    return RT_OK;
}

int rt_get_counters(RtContext *c, RtCounters *out) {
    if (!c || !out) return RT_ERR_INVALID;
    if (!c->cfg.countWork) return fail(c, RT_ERR_STATE, "rt_get_counters: context created without countWork");
    (void)hipSetDevice(c->cfg.device);
    unsigned long long v[16];
    HIP_TRY(c, sync_all(c));
    HIP_TRY(c, hipMemcpy(v, c->dCounters, sizeof v, hipMemcpyDeviceToHost));
    out->raysClosest = v[0]; out->raysShadow = v[1]; out->raysAnalytic = v[2]; out->nodeFetch = v[3];
    out->triFetch = v[4]; out->envLookup = v[5]; out->hitPixels = v[6];
    out->fetchPrimary = v[7]; out->fetchShadow = v[8]; out->fetchAO = v[9];
    return RT_OK;
}
int rt_reset_counters(RtContext *c) {
    if (!c) return RT_ERR_INVALID;
    (void)hipSetDevice(c->cfg.device);
    // frames still in flight on ANY lane add their atomics until they finish: drain all lanes, then clear synchronously
    HIP_TRY(c, sync_all(c));
    HIP_TRY(c, hipMemset(c->dCounters, 0, 16 * sizeof(unsigned long long)));
    return RT_OK;
}

int rt_get_scene_info(const RtContext *c, RtSceneInfo *out) {
    if (!c || !out) return RT_ERR_INVALID;
    std::memset(out, 0, sizeof *out);
    out->nNodes = c->nNodes; out->nTris = c->nTris; out->nInner = c->nInner; out->treeDepth = c->treeDepth;
    if (c->nNodes == 0) return RT_OK;
    out->nWide4 = (int32_t)c->nWide4; out->nPairs = (int32_t)c->nPairs;
    out->bytesNodes2 = (uint64_t)std::max(c->nInner, 1) * 64;
    out->bytesNodes4 = c->dQ4 ? (uint64_t)c->nWide4 * 64 + (uint64_t)c->nLeafBoxes * 32 : (uint64_t)c->nWide4 * 128;   // the any-hit launches walk the quantised nodes when they exist
    out->bytesPairs = (uint64_t)c->nPairs * 80;
    out->bytesTris = (uint64_t)c->nTris * 48;
    out->nFused = (int32_t)c->nFused;
    out->flags = c->sceneFlags | (c->dIN2 ? RT_SCENE_IMPLICIT : 0);
    out->implicitDepth = c->implD;
    return RT_OK;
}

int rt_get_memory_info(RtContext *c, RtMemoryInfo *out) {
    if (!c || !out) return RT_ERR_INVALID;
    std::memset(out, 0, sizeof *out);
    (void)hipSetDevice(c->cfg.device);
    out->queueArenaBytes = rt_arena_pool_bytes(c->arenaPool);
    out->queueArenas = rt_arena_pool_count(c->arenaPool);
    out->lanes = c->nLanes;
    for (int i = 0; i < c->nLanes; ++i) { out->frameArrayBytes += rt_wave_frame_bytes(c->wave[i]); out->hybridArenaBytes += rt_hybrid_arena_bytes(c->hybrid[i]); }
    size_t fr = 0, tot = 0;
    HIP_TRY(c, hipMemGetInfo(&fr, &tot));
    out->deviceFreeBytes = fr; out->deviceTotalBytes = tot;
    return RT_OK;
}

int rt_get_traced_rays(RtContext *c, RtTracedRays *out, int reset) {
    if (!c || !out) return RT_ERR_INVALID;
    (void)hipSetDevice(c->cfg.device);
    unsigned long long v[16] = {0};
    (void)sync_all(c);
    for (int i = 0; i < c->nLanes; ++i) {
        unsigned long long t[16];
        int rc = rt_wave_traced(c->wave[i], c->lanes[i], t, reset != 0);
        if (rc != RT_OK) return fail(c, rc, "rt_get_traced_rays: %s", rt_wave_error(c->wave[i]));
        for (int k = 0; k < 16; ++k) v[k] += t[k];
    }
    out->candidatePixels = v[0]; out->hitPixels = v[1]; out->primary = v[2]; out->shadow = v[3]; out->bounce = v[4];
    out->bounceShadow = v[5]; out->frames = v[6];
    out->gatherLoadsPrimary = v[8]; out->gatherLoadsShadow = v[9]; out->gatherLoadsBounce = v[10];
    out->mergedLoadsPrimary = v[11]; out->mergedLoadsShadow = v[12]; out->mergedLoadsBounce = v[13];
    out->ao = v[7]; out->gatherLoadsAO = v[14];
    return RT_OK;
}

int rt_enable_stage_timing(RtContext *c, int enable) {
    if (!c) return RT_ERR_INVALID;
    (void)hipSetDevice(c->cfg.device);
    resolve_stage_events(c);
    c->timing = enable != 0;
    std::memset(c->stageMs, 0, sizeof c->stageMs);
    std::memset(c->stageLaunches, 0, sizeof c->stageLaunches);
    c->timedFrames = 0;
    return RT_OK;
}
int rt_get_stage_times(RtContext *c, RtStageTimes *out) {
    if (!c || !out) return RT_ERR_INVALID;
    (void)hipSetDevice(c->cfg.device);
    resolve_stage_events(c);
    out->nStages = RT_MAX_STAGES;
    out->frames = c->timedFrames;
    for (int i = 0; i < RT_MAX_STAGES; ++i) { out->ms[i] = c->stageMs[i]; out->launches[i] = c->stageLaunches[i]; }
    return RT_OK;
}

int rt_debug_eval(RtContext *c, int op, const float *a, const float *b, const float *cc, uint32_t *out, int n) {
    if (!c || !a || !out || n <= 0) return RT_ERR_INVALID;
    (void)hipSetDevice(c->cfg.device);
    float *da = nullptr, *db = nullptr, *dc = nullptr;
    uint32_t *dout = nullptr;
    const size_t bytes = (size_t)n * 4;
    HIP_TRY(c, hipMalloc(&da, bytes));
    HIP_TRY(c, hipMalloc(&dout, bytes));
    HIP_TRY(c, hipMemcpy(da, a, bytes, hipMemcpyHostToDevice));
    if (b) { HIP_TRY(c, hipMalloc(&db, bytes)); HIP_TRY(c, hipMemcpy(db, b, bytes, hipMemcpyHostToDevice)); }
    if (cc) { HIP_TRY(c, hipMalloc(&dc, bytes)); HIP_TRY(c, hipMemcpy(dc, cc, bytes, hipMemcpyHostToDevice)); }
    hipLaunchKernelGGL(k_debug_eval, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, op, da, db, dc, dout, n);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, sync_all(c));
    HIP_TRY(c, hipMemcpy(out, dout, bytes, hipMemcpyDeviceToHost));
    (void)hipFree(da); (void)hipFree(dout);
    if (db) (void)hipFree(db);
    if (dc) (void)hipFree(dc);
    return RT_OK;
}

// kinds 2 / 3: the same rays through the wavefront pipeline's traversal kernels (k_trace, as the frames launch it)
static int debug_trace_wave(RtContext *c, bool any, const float *origins, const float *dirs, const float *tMax, float eps, float inf, float *out7, int n) {
    if (c->nNodes <= 0 || c->nTris <= 0) return fail(c, RT_ERR_INVALID, "rt_debug_trace: no BVH uploaded");
    std::vector<float> o4((size_t)n * 4, 0.0f), d4((size_t)n * 4, 0.0f), tm((size_t)n, inf);
    for (int i = 0; i < n; ++i) {
        std::memcpy(&o4[(size_t)i * 4], origins + (size_t)i * 3, 12);
        std::memcpy(&d4[(size_t)i * 4], dirs + (size_t)i * 3, 12);
        if (any) tm[(size_t)i] = tMax[i];
    }
    std::vector<unsigned char> hostBytes(sizeof(DevFrame), 0);
    DevFrame *host = reinterpret_cast<DevFrame *>(hostBytes.data());
    host->u.eps = eps; host->u.inf = inf;
    host->sc = make_dev_scene(c);
    host->giBounces = 1;
    float4 *dO = nullptr, *dD = nullptr;
    float *dT = nullptr, *dOutT = nullptr;
    int *dTri = nullptr;
    uint8_t *dOcc = nullptr;
    uint32_t *dCnt = nullptr, *dHeads = nullptr;
    DevFrame *dF = nullptr;
    auto freeAll = [&]() { for (void *p : {(void *)dO, (void *)dD, (void *)dT, (void *)dOutT, (void *)dTri, (void *)dOcc, (void *)dCnt, (void *)dHeads, (void *)dF}) if (p) (void)hipFree(p); };
    bool ok = hipMalloc(&dO, (size_t)n * 16) == hipSuccess && hipMalloc(&dD, (size_t)n * 16) == hipSuccess && hipMalloc(&dT, (size_t)n * 4) == hipSuccess &&
              hipMalloc(&dOutT, (size_t)n * 4) == hipSuccess && hipMalloc(&dTri, (size_t)n * 4) == hipSuccess && hipMalloc(&dOcc, (size_t)n) == hipSuccess &&
              hipMalloc(&dCnt, 4) == hipSuccess && hipMalloc(&dHeads, rt_wave_head_words() * 4) == hipSuccess && hipMalloc(&dF, sizeof(DevFrame)) == hipSuccess;
    const uint32_t un = (uint32_t)n;
    ok = ok && hipMemcpy(dO, o4.data(), (size_t)n * 16, hipMemcpyHostToDevice) == hipSuccess && hipMemcpy(dD, d4.data(), (size_t)n * 16, hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(dT, tm.data(), (size_t)n * 4, hipMemcpyHostToDevice) == hipSuccess && hipMemcpy(dCnt, &un, 4, hipMemcpyHostToDevice) == hipSuccess &&
         hipMemset(dHeads, 0, rt_wave_head_words() * 4) == hipSuccess && hipMemset(dOcc, 0, (size_t)n) == hipSuccess && hipMemset(dTri, 0xff, (size_t)n * 4) == hipSuccess &&
         hipMemcpy(dF, host, sizeof(DevFrame), hipMemcpyHostToDevice) == hipSuccess;
    if (!ok) { freeAll(); return fail(c, RT_ERR_HIP, "rt_debug_trace: allocation / upload failed"); }
    rt_wave_debug_trace(c->stream, c->cus, c->treeDepth, dF, host->sc, any, dO, dD, dT, dCnt, un, dOutT, dTri, dOcc, dHeads);
    std::vector<float> t((size_t)n);
    std::vector<int> tri((size_t)n);
    std::vector<uint8_t> occ((size_t)n);
    ok = hipGetLastError() == hipSuccess && sync_all(c) == hipSuccess && hipMemcpy(t.data(), dOutT, (size_t)n * 4, hipMemcpyDeviceToHost) == hipSuccess &&
         hipMemcpy(tri.data(), dTri, (size_t)n * 4, hipMemcpyDeviceToHost) == hipSuccess && hipMemcpy(occ.data(), dOcc, (size_t)n, hipMemcpyDeviceToHost) == hipSuccess;
    freeAll();
    if (!ok) return fail(c, RT_ERR_HIP, "rt_debug_trace: launch failed");
    for (int i = 0; i < n; ++i) {
        float *out = out7 + (size_t)i * 7;
        for (int k = 0; k < 7; ++k) out[k] = 0.0f;
        if (any) out[0] = occ[(size_t)i] ? 1.0f : 0.0f;
        else { out[0] = tri[(size_t)i] >= 0 ? t[(size_t)i] : inf; out[1] = (float)tri[(size_t)i]; }   // closest: t and the triangle's index in the reference order
    }
    return RT_OK;
}

int rt_debug_trace(RtContext *c, int kind, const float *origins, const float *dirs, const float *tMax, float eps, float inf, float *out7, int n) {
    if (!c || !origins || !dirs || !out7 || n <= 0 || ((kind == 1 || kind == 3) && !tMax) || kind < 0 || kind > 3) return RT_ERR_INVALID;
    (void)hipSetDevice(c->cfg.device);
    if (kind >= 2) return guarded(c, "rt_debug_trace", [&]() -> int { return debug_trace_wave(c, kind == 3, origins, dirs, tMax, eps, inf, out7, n); });
    float *dO = nullptr, *dD = nullptr, *dT = nullptr, *dOut = nullptr;
    HIP_TRY(c, hipMalloc(&dO, (size_t)n * 12));
    HIP_TRY(c, hipMalloc(&dD, (size_t)n * 12));
    HIP_TRY(c, hipMalloc(&dT, (size_t)n * 4));
    HIP_TRY(c, hipMalloc(&dOut, (size_t)n * 28));
    HIP_TRY(c, hipMemcpy(dO, origins, (size_t)n * 12, hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(dD, dirs, (size_t)n * 12, hipMemcpyHostToDevice));
    if (tMax) HIP_TRY(c, hipMemcpy(dT, tMax, (size_t)n * 4, hipMemcpyHostToDevice));
    DevScene sc = make_dev_scene(c);
    hipLaunchKernelGGL(k_debug_trace, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, sc, kind, dO, dD, dT, eps, inf, dOut, n);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, sync_all(c));
    HIP_TRY(c, hipMemcpy(out7, dOut, (size_t)n * 28, hipMemcpyDeviceToHost));
    (void)hipFree(dO); (void)hipFree(dD); (void)hipFree(dT); (void)hipFree(dOut);
    return RT_OK;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------
// Tile-parallel exchange over RCCL, owned by the library.  librccl is half a gigabyte and only multi-GPU
// runs need it, so it is bound on first use: an already loaded copy (e.g. the one a PyTorch wheel brought into the process) is
// reused by SONAME, otherwise librccl.so.1 is loaded from the ROCm installation.
namespace {
struct RcclApi {
    decltype(&ncclGetUniqueId) getUniqueId = nullptr;
    decltype(&ncclCommInitRank) commInitRank = nullptr;
    decltype(&ncclCommDestroy) commDestroy = nullptr;
    decltype(&ncclGroupStart) groupStart = nullptr;
    decltype(&ncclGroupEnd) groupEnd = nullptr;
    decltype(&ncclSend) send = nullptr;
    decltype(&ncclRecv) recv = nullptr;
    decltype(&ncclAllGather) allGather = nullptr;
    decltype(&ncclGetErrorString) errorString = nullptr;
    decltype(&ncclCommCount) commCount = nullptr;          // optional (rt_comm_info)
    decltype(&ncclCommUserRank) commUserRank = nullptr;
    std::string err;
    bool ok = false;
};
static void rccl_bind(RcclApi &a);
RcclApi &rccl_api() {
    static RcclApi a;
    static std::once_flag once;              // contexts on different threads may reach their first rt_comm_* call together
    std::call_once(once, [] { rccl_bind(a); });
    return a;
}
static void rccl_bind(RcclApi &a) {
    void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD);
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) { const char *e = dlerror(); a.err = std::string("librccl.so.1 could not be loaded: ") + (e ? e : "?"); return; }
#define RT_BIND(field, name) do { a.field = (decltype(a.field))dlsym(h, #name); if (!a.field) { a.err = "librccl lacks " #name; return; } } while (0)
    RT_BIND(getUniqueId, ncclGetUniqueId); RT_BIND(commInitRank, ncclCommInitRank); RT_BIND(commDestroy, ncclCommDestroy);
    RT_BIND(groupStart, ncclGroupStart); RT_BIND(groupEnd, ncclGroupEnd); RT_BIND(send, ncclSend); RT_BIND(recv, ncclRecv);
    RT_BIND(allGather, ncclAllGather); RT_BIND(errorString, ncclGetErrorString);
#undef RT_BIND
    a.commCount = (decltype(a.commCount))dlsym(h, "ncclCommCount");
    a.commUserRank = (decltype(a.commUserRank))dlsym(h, "ncclCommUserRank");
    a.ok = true;
}
#define NCCL_TRY(c, expr)                                                                                           \
    do {                                                                                                            \
        ncclResult_t r_ = (expr);                                                                                   \
        if (r_ != ncclSuccess) return fail((c), RT_ERR_HIP, "%s failed: %s", #expr, rccl_api().errorString(r_));     \
    } while (0)
void *lane_target(RtContext *c, int lane, int which, int &ch) {
    switch (which) {
        case RT_TARGET_COLOR: ch = 4; return c->dColor[lane];
        case RT_TARGET_MOTION: ch = 2; return c->dMotion[lane];
        case RT_TARGET_GPOS: ch = 4; return c->dGPos[lane];
        case RT_TARGET_GNRM: ch = 4; return c->dGNrm[lane];
        default: ch = 0; return nullptr;
    }
}
}  // namespace

extern "C" {

int rt_comm_unique_id(void *id, size_t bytes) {
    if (!id || bytes < RT_COMM_ID_BYTES) return fail(nullptr, RT_ERR_INVALID, "rt_comm_unique_id: need %d bytes", RT_COMM_ID_BYTES);
    RcclApi &a = rccl_api();
    if (!a.ok) return fail(nullptr, RT_ERR_UNSUPPORTED, "rt_comm_unique_id: %s", a.err.c_str());
    static_assert(sizeof(ncclUniqueId) == RT_COMM_ID_BYTES, "RT_COMM_ID_BYTES");
    ncclUniqueId u;
    ncclResult_t r = a.getUniqueId(&u);
    if (r != ncclSuccess) return fail(nullptr, RT_ERR_HIP, "ncclGetUniqueId: %s", a.errorString(r));
    std::memcpy(id, &u, sizeof u);
    return RT_OK;
}

int rt_comm_init(RtContext *c, const void *id, size_t bytes) {
    if (!c || !id || bytes < RT_COMM_ID_BYTES) return RT_ERR_INVALID;
    if (c->comm) return fail(c, RT_ERR_STATE, "rt_comm_init: the context already has a communicator");
    RcclApi &a = rccl_api();
    if (!a.ok) return fail(c, RT_ERR_UNSUPPORTED, "rt_comm_init: %s", a.err.c_str());
    (void)hipSetDevice(c->cfg.device);
    HIP_TRY(c, sync_all(c));
    ncclUniqueId u;
    std::memcpy(&u, id, sizeof u);
    ncclComm_t comm = nullptr;
    NCCL_TRY(c, a.commInitRank(&comm, c->cfg.worldSize, u, c->cfg.rank));   // collective over all ranks of the frame
    c->comm = (void *)comm;
    // the gathering rank's COLOR0 buffers for every frame lane now, not inside the frame loop (a first-gather hipMalloc would otherwise land in
    // whatever is being timed; the other targets are gathered for a present only and keep allocating on first use)
    if (c->sized && c->g.rank == 0) {
        const size_t block = c->nSlots * 8;
        for (int l = 0; l < c->nLanes; ++l) {
            if (!c->dGathered[l][RT_TARGET_COLOR]) HIP_TRY(c, hipMalloc(&c->dGathered[l][RT_TARGET_COLOR], block * (size_t)c->g.world));
            if (!c->dAssembled[l][RT_TARGET_COLOR]) HIP_TRY(c, hipMalloc(&c->dAssembled[l][RT_TARGET_COLOR], (size_t)c->g.W * c->g.H * 8));
        }
    }
    return RT_OK;
}

int rt_comm_destroy(RtContext *c) {
    if (!c) return RT_ERR_INVALID;
    if (!c->comm) return RT_OK;
    (void)hipSetDevice(c->cfg.device);
    (void)sync_all(c);
    ncclResult_t r = rccl_api().commDestroy((ncclComm_t)c->comm);
    c->comm = nullptr;
    if (r != ncclSuccess) return fail(c, RT_ERR_HIP, "ncclCommDestroy: %s", rccl_api().errorString(r));
    return RT_OK;
}

// One exchange per gathered frame: every rank sends the block of its tiles to rank 0 -- grouped point-to-point
// transfers, so the root's inbound xGMI links run in parallel and nothing is reduced -- and rank 0 un-tiles the blocks into a
// row-major frame.  Everything is enqueued on the lane (stream) of the frame rendered last and uses that lane's own buffers, so
// gathers of consecutive frames overlap like the frames themselves and never share a buffer.
int rt_gather_frame(RtContext *c, int which) {
    if (!c) return RT_ERR_INVALID;
    if (!c->sized) return fail(c, RT_ERR_STATE, "rt_gather_frame before rt_resize");
    if (c->frameIndex == 0) return fail(c, RT_ERR_STATE, "rt_gather_frame before the first frame");
    if (c->g.world > 1 && !c->comm) return fail(c, RT_ERR_STATE, "rt_gather_frame on a tile-parallel context without rt_comm_init");
    (void)hipSetDevice(c->cfg.device);
    const int lane = last_lane(c);
    int ch;
    void *local = lane_target(c, lane, which, ch);
    if (!local) return fail(c, RT_ERR_INVALID, "rt_gather_frame: which=%d", which);
    const size_t block = c->nSlots * (size_t)ch * 2;
    hipStream_t st = c->lanes[lane];
    const bool root = c->g.rank == 0;
    if (root) {
        if (!c->dGathered[lane][which]) HIP_TRY(c, hipMalloc(&c->dGathered[lane][which], block * (size_t)c->g.world));
        if (!c->dAssembled[lane][which]) HIP_TRY(c, hipMalloc(&c->dAssembled[lane][which], (size_t)c->g.W * c->g.H * ch * 2));
    }
    // stage "gather": from the point the lane's stream reaches the exchange (its frame is done) to the end of the un-tiling on the root / of
    // the send on the others -- on the root this is what the first 8-GPU run needs to see next to the ranks' frame times (bench.py)
    struct GatherSpan {
        RtContext *c; hipStream_t st;
        GatherSpan(RtContext *c_, hipStream_t s_) : c(c_), st(s_) { rt_stage_begin(c, 12, st); }
        ~GatherSpan() { rt_stage_end(c, 12, 1, st); }
    } span(c, st);
    c->gathers++;
    c->gatherBytes += root ? block * (size_t)(c->g.world - 1) : block;
    if (root) HIP_TRY(c, hipMemcpyAsync(c->dGathered[lane][which], local, block, hipMemcpyDeviceToDevice, st));
    if (c->g.world > 1) {
        RcclApi &a = rccl_api();
        ncclComm_t comm = (ncclComm_t)c->comm;
        NCCL_TRY(c, a.groupStart());
        // a failing send / recv must not leave the communicator inside an open group (later collectives and ncclCommDestroy would
        // hang): close the group first, then report the first error
        ncclResult_t r1 = ncclSuccess;
        if (root) {
            for (int r = 1; r < c->g.world && r1 == ncclSuccess; ++r)
                r1 = a.recv((char *)c->dGathered[lane][which] + (size_t)r * block, block, ncclUint8, r, comm, st);
        } else {
            r1 = a.send(local, block, ncclUint8, 0, comm, st);
        }
        const ncclResult_t r2 = a.groupEnd();
        if (r1 != ncclSuccess) return fail(c, RT_ERR_HIP, "ncclSend/ncclRecv failed: %s", a.errorString(r1));
        if (r2 != ncclSuccess) return fail(c, RT_ERR_HIP, "ncclGroupEnd failed: %s", a.errorString(r2));
    }
    if (root) {
        const size_t n = (size_t)c->g.W * c->g.H;
        rt_stage_begin(c, 10, st);
        hipLaunchKernelGGL(k_assemble, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const void *)c->dGathered[lane][which],
                           c->dAssembled[lane][which], c->g, ch, block);
        rt_stage_end(c, 10, 1, st);
        HIP_TRY(c, hipGetLastError());
    }
    c->gatheredLane[which] = lane;
    return RT_OK;
}

int rt_gathered_frame(RtContext *c, int which, void **devPtr, size_t *bytes) {
    if (!c || !devPtr || !bytes || which < 0 || which > 3) return RT_ERR_INVALID;
    if (c->g.rank != 0) return fail(c, RT_ERR_STATE, "rt_gathered_frame: only rank 0 holds the gathered frame");
    const int lane = c->gatheredLane[which];
    if (lane < 0 || !c->dAssembled[lane][which]) return fail(c, RT_ERR_STATE, "rt_gathered_frame: no rt_gather_frame(%d) since the last reset", which);
    *devPtr = c->dAssembled[lane][which];
    *bytes = (size_t)c->g.W * c->g.H * (which == RT_TARGET_MOTION ? 2 : 4) * 2;
    return RT_OK;
}

int rt_read_gathered(RtContext *c, int which, void *dstHalfs) {
    if (!c || !dstHalfs) return RT_ERR_INVALID;
    void *p;
    size_t n;
    int rc = rt_gathered_frame(c, which, &p, &n);
    if (rc != RT_OK) return rc;
    (void)hipSetDevice(c->cfg.device);
    HIP_TRY(c, sync_all(c));
    HIP_TRY(c, hipMemcpy(dstHalfs, p, n, hipMemcpyDeviceToHost));
    return RT_OK;
}

int rt_present_last_gathered(RtContext *c, const RtPresentParams *p, uint8_t *dst) {
    if (!c || !p || !dst) return RT_ERR_INVALID;
    if (c->g.rank != 0) return fail(c, RT_ERR_STATE, "rt_present_last_gathered: only rank 0 holds the gathered targets");
    const int lane = c->gatheredLane[0];
    for (int w = 0; w < 4; ++w)
        if (c->gatheredLane[w] < 0 || c->gatheredLane[w] != lane || !c->dGathered[lane][w])
            return fail(c, RT_ERR_STATE, "rt_present_last_gathered: gather all four targets of the same frame first (rt_gather_frame 0..3)");
    return rt_present_gathered(c, p, c->dGathered[lane][0], c->dGathered[lane][1], c->dGathered[lane][2], c->dGathered[lane][3], dst);
}

// Moving camera on a tile-parallel frame: every rank needs the whole previous COLOR0 (rt_taa.glsl:116-179 reads it at arbitrary
// pixels) -> one all-gather of the ranks' blocks into this lane's exchange buffer, then the event the next frame's resolve waits on.
int rt_exchange_history(RtContext *c) {
    if (!c) return RT_ERR_INVALID;
    if (!c->sized) return fail(c, RT_ERR_STATE, "rt_exchange_history before rt_resize");
    if (c->g.world > 1 && !c->comm) return fail(c, RT_ERR_STATE, "rt_exchange_history on a tile-parallel context without rt_comm_init");
    void *buf;
    size_t bytes;
    int rc = rt_history_exchange_buffer(c, &buf, &bytes);
    if (rc != RT_OK) return rc;
    const int lane = last_lane(c);
    const size_t block = c->nSlots * 8;
    if (c->g.world > 1) NCCL_TRY(c, rccl_api().allGather(c->dColor[lane], buf, block, ncclUint8, (ncclComm_t)c->comm, c->lanes[lane]));
    else HIP_TRY(c, hipMemcpyAsync(buf, c->dColor[lane], block, hipMemcpyDeviceToDevice, c->lanes[lane]));
    c->historyExchanges++;
    return rt_history_exchanged(c);
}

int rt_comm_info(RtContext *c, RtCommInfo *out) {
    if (!c || !out) return RT_ERR_INVALID;
    out->commWorld = out->commRank = -1;
    out->rank = c->cfg.rank; out->worldSize = c->cfg.worldSize;
    out->gathers = c->gathers; out->gatherBytes = c->gatherBytes; out->historyExchanges = c->historyExchanges;
    if (c->comm) {
        RcclApi &a = rccl_api();
        int v = -1;
        if (a.commCount && a.commCount((ncclComm_t)c->comm, &v) == ncclSuccess) out->commWorld = v;
        v = -1;
        if (a.commUserRank && a.commUserRank((ncclComm_t)c->comm, &v) == ncclSuccess) out->commRank = v;
    }
    return RT_OK;
}

}  // extern "C"

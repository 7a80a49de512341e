// rt_wave.hip -- wavefront pipeline for BVH scenes (the production path on MI355X).
//
// One frame of rt.frag in BVH mode is cut into stages so that every traversal runs in a dedicated,
// register-lean persistent kernel and all shading runs in dense ALU kernels:
//
//   primary        thread = pixel.  Ray generation + slab test against the root box.  Pixels whose
//                  ray misses the scene box (the vast majority when the mesh is small on screen)
//                  are finished here: sky x SPP, TAA resolve, 4 target stores.  The rest are
//                  compacted (wave ballot, one atomic per wave) into the candidate list.
//   trace_primary  persistent closest-hit traversal over the candidates (ray rebuilt from the pixel).
//   post_primary   thread = candidate.  Misses are finished like above; hits are compacted into
//                  the hit list {slot, t, triangle}.
//   gen_direct     thread = (hit, sample).  Runs the reference's shading code with a tracer that
//                  only RECORDS rays: 4 disk + sun + point shadow rays, the GI bounce ray, and
//                  (sample 0) the AO rays.  Queues are slot-major: all "sun" rays of a chunk are
//                  contiguous, so neighbouring lanes trace near-identical rays.
//   trace_shadow   persistent any-hit traversal (direct shadows + AO as any-hit with tMax < radius)
//   trace_gi       persistent closest-hit traversal of the bounce rays
//   gen_gi         thread = (hit, sample): shadow rays at the bounce hit
//   trace_gi_shadow
//   combine        thread = hit: the reference's shading code again, now with a tracer that READS
//                  the recorded visibilities / hits, in the reference's evaluation order (so the
//                  sums are bit-identical to the megakernel and the oracle); TAA; stores.
//
// Persistent traversal ("ray scheduler"): a fixed grid of waves pulls rays from a global cursor.
// When enough lanes of a wave have finished their ray (ballot + popcount), exactly those lanes
// fetch new rays with ONE atomic for the wave and restart, so divergent ray lengths do not leave
// the SIMD half empty.  Each lane's traversal stack is in LDS (stack[e*64 + lane], 8-byte entries:
// deferred child + its entry distance).
#include "rt_wave.hpp"

#include <algorithm>
#include <cstdio>
#include <map>
#include <string>
#include <tuple>
#include <vector>

#include "../../include/rt_mi355.h"

#pragma clang fp contract(off)

using namespace rtd;

// stage ids (rt_stage_name in rt_api.hip)
enum { ST_PRIMARY = 1, ST_TRACE_PRIMARY, ST_POST_PRIMARY, ST_GEN_DIRECT, ST_TRACE_SHADOW, ST_TRACE_GI, ST_GEN_GI, ST_RESOLVE, ST_COMBINE, ST_TRACE_AO = 13 };

struct HitRec { uint32_t slot; float t; int tri; };

struct WaveBuf {
    // per frame
    uint32_t *cand;          // candidate pixel slots
    uint32_t *counts;        // [0] candidates, [1] hits, [2..] traced-ray tallies
    uint32_t *heads;         // ray cursors, one per trace launch
    float *primT;            // per candidate
    int *primTri;
    HitRec *hits;
    // per chunk of CH hits
    float4 *shO, *shD;       // shadow queue 1: (A + 6*SPP) slots x CH
    float *shT, *giL, *sh2T; // per-slot tMax (any-hit) / liveness (bounce); < 0 = no ray in this slot (4 B instead of a 32-B record)
    uint8_t *occ1;
    float4 *giO, *giD;       // bounce queue: SPP slots x CH
    float *giT;
    int *giTri;
    float4 *sh2O, *sh2D;     // shadow queue 2: 6 slots x q2Stride, entries compacted over the (hit, sample) pairs whose bounce hit
    uint32_t q2Stride;       // entries per slot of queue 2: CH * SPP (every bounce ray may hit) for small launch sets; for large ones (round 5) a capacity PREDICTED from the bounce
                             // hits of earlier batches -- a (hit, sample) pair whose entry lies beyond it is not queued: k_gen_gi_overflow traces its six rays in place
    uint8_t *occOvf;         // answers of those rays, [6][CH * SPP] (per lane, like occ2)
    uint8_t *occ2;
    int *giPos;              // per (sample, hit): entry in queue 2, -1 when the bounce ray missed or was not cast
    int *giPerm;             // RT_BIN_GI=1 (experiment, round 4): per (sample, hit) the bounce queue entry its ray was sorted to; null = entry (sample, hit) itself
    // per frame, per pixel slot: everything the frame produced BEFORE the temporal resolve (the only history-dependent step)
    float4 *pendC;           // curr.rgb (frame average, fp32), motion.x
    float *pendMy;           // motion.y
    uint2 *pendPos, *pendNrm;
    uint32_t CH;             // chunk capacity (hits)
    int A;                   // AO rays per hit (0 when AO is off)
    int SPP;
    // slot of shadow queue 1 for ray k of sample s: A AO slots, then the four disk-light rays of every sample, then ONE sun and ONE point-light
    // slot per hit -- those two rays do not depend on the sample (rt_lighting.glsl:114-214), sample 0 traces them and the others reuse its answer,
    // so samples > 0 own no slot for them (round 4: 22 instead of 28 slots per hit at 4 spp)
    __device__ __forceinline__ uint32_t gi_entry(int s, uint32_t j) const { const uint32_t a = (uint32_t)s * CH + j; return giPerm ? (uint32_t)giPerm[a] : a; }
    __device__ __forceinline__ uint32_t sh1_slot(int s, int k) const { return (uint32_t)(k < 4 ? A + s * 4 + k : A + 4 * SPP + (k - 4)); }
};

namespace {


// ---- finishing a pixel.  rt.frag:184-196 = frame average -> TAA resolve against the history -> 4 target stores.  The history
// read is the ONLY dependency of a frame on its predecessor, so the stages stash the pre-resolve values and a final k_resolve
// does TAA + stores.  Frames f and f+1 run on two streams and overlap everywhere except resolve(f) -> resolve(f+1), which
// keeps the GPU busy when one frame's stages are latency-bound (tile-parallel ranks with 1/8 of the pixels, small frames).
RT_DEV void finish_pixel(const DevFrame *fr, const WaveBuf &wb, int slot, V3 frameSum, V2 motionOut, V4 gpos, V4 gnrm) {
    const int SPP = max(fr->u.spp, 1);
    V3 curr = frameSum / (float)SPP;
    wb.pendC[slot] = make_float4(curr.x, curr.y, curr.z, motionOut.x);
    wb.pendMy[slot] = motionOut.y;
    wb.pendPos[slot] = pack_half4(gpos);
    wb.pendNrm[slot] = pack_half4(gnrm);
}
// History of the still-camera resolve inside a batch: frame k > 0 reads what frame k-1 of the same batch just produced, rounded to
// fp16 as the RGBA16F target would hold it.
struct ChainedHistory {
    HistoryTex tex;
    bool chained;
    V4 value;
    RT_DEV V4 own() const { return chained ? value : tex.own(); }
    RT_DEV V4 at(float u, float v) const { return tex.at(u, v); }   // reprojection: never inside a batch (static camera only)
};
__global__ __launch_bounds__(256) void k_resolve(const DevFrame *__restrict__ fr, Targets tg, WaveBuf wb) {
    const RtUniforms &u = fr->u;
    int px, py;
    if (!pixel_of_slot(fr->g, blockIdx.x, threadIdx.x, px, py)) return;
    const int slot = blockIdx.x * 256 + threadIdx.x;
    float uvx = ((float)px + 0.5f) / (float)fr->g.W, uvy = ((float)py + 0.5f) / (float)fr->g.H;   // rt_fullscreen.vert:44
    ChainedHistory hist;
    hist.tex.prev = tg.prev; hist.tex.prevAll = tg.prevAll; hist.tex.blockSlots = tg.blockSlots; hist.tex.g = &fr->g; hist.tex.slot = slot;
    hist.chained = false;
    const int K = max(fr->g.batch, 1);
    const int perFrame = fr->g.nLocalTiles * 256;   // slots of one frame of the batch
    uint2 colorBits = make_uint2(0u, 0u);
    V2 motionOut = mk2(0.0f, 0.0f);
    int sk = slot;
    for (int k = 0; k < K; ++k, sk += perFrame) {
        float4 pc = wb.pendC[sk];
        V3 curr = mk3(pc.x, pc.y, pc.z);
        motionOut = mk2(pc.w, wb.pendMy[sk]);
        V2 taaMotion = (u.cameraMoved == 1) ? motionOut : mk2(0.0f, 0.0f);
        V4 taa = resolveTAA(u, curr, uvx, uvy, taaMotion, u.frameIndex + k, hist);
        colorBits = pack_half4(taa);
        hist.chained = true;
        hist.value = unpack_half4(colorBits);        // what the next frame's texture(uPrevAccum, uv) returns: the fp16 target
    }
    sk -= perFrame;                                  // the targets hold the batch's last frame
    tg.color[slot] = colorBits;
    tg.motion[slot] = pack_half2(motionOut);
    tg.gpos[slot] = wb.pendPos[sk];
    tg.gnrm[slot] = wb.pendNrm[sk];
}
RT_DEV void finish_miss(const DevFrame *fr, const WaveBuf &wb, int slot, int px, int py, V3 dir) {
    const RtUniforms &u = fr->u;
    Frag F;
    F.u = &u; F.sc = &fr->sc; F.fcx = (float)px + 0.5f; F.fcy = (float)py + 0.5f;
    F.frameIndex = u.frameIndex + sub_frame_of_slot(fr->g, (uint32_t)slot);
    Work w;
    V3 r = sky<false>(F, dir, w);
    V3 frameSum = mk3(0.0f);
    const int SPP = max(u.spp, 1);
    for (int s = 0; s < SPP; ++s) frameSum = frameSum + r;   // the reference adds the same radiance SPP times
    V2 motionOut = (u.cameraMoved == 1) ? mk2(4.0f, 4.0f) : mk2(0.0f, 0.0f);
    finish_pixel(fr, wb, slot, frameSum, motionOut, mk4(0, 0, 0, 0), mk4(0, 0, 0, 0));
}
RT_DEV void slot_to_pixel(const FrameGeom &g, uint32_t slot, int &px, int &py) { pixel_of_slot(g, (int)(slot >> 8), (int)(slot & 255u), px, py); }
// primary ray of pixel (px, py) in the batch's k-th frame: that frame's jitter (rt.frag:58-68)
RT_DEV V3 primaryDirK(const DevFrame *fr, int k, int px, int py) { return primaryDirJ(fr->u, (float)px + 0.5f, (float)py + 0.5f, fr->jitterK[k][0], fr->jitterK[k][1]); }

// wave-level append: returns this lane's index in the list (valid where pred)
RT_DEV uint32_t wave_append(bool pred, uint32_t *counter) {
    unsigned long long m = __ballot(pred);
    uint32_t n = (uint32_t)__popcll(m);
    uint32_t base = 0;
    if (n) {
        int leader = __ffsll((long long)m) - 1;
        if ((int)(threadIdx.x & 63) == leader) base = atomicAdd(counter, n);
        base = __shfl(base, leader, 64);
    }
    uint32_t lane = threadIdx.x & 63;
    uint32_t rank = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    return base + rank;
}

// block-level append (all 256 threads must call): ONE atomic per workgroup on the list counter -- a single
// counter word sustains only ~88 M atomics/s, which per-wave appends of a 1080p frame would saturate.
RT_DEV uint32_t block_append(bool pred, uint32_t *counter) {
    __shared__ uint32_t s_cnt[4];
    __shared__ uint32_t s_base;
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    unsigned long long m = __ballot(pred);
    if (lane == 0) s_cnt[wv] = (uint32_t)__popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t tot = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
        s_base = tot ? atomicAdd(counter, tot) : 0u;
    }
    __syncthreads();
    uint32_t off = s_base;
    for (uint32_t i = 0; i < wv; ++i) off += s_cnt[i];
    uint32_t r = off + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    __syncthreads();   // s_cnt / s_base may be reused by a second append in the same kernel
    return r;
}

// The same for kAppendBatch sub-blocks of 256 items handled by one workgroup: still ONE atomic, for 2048 items.  (With one
// atomic per 256 items k_primary spent 0.10 ms of a 1080p frame queueing 8100 atomics on one word; now 0.025 ms.)
//   note(k, pred) for every sub-block k, commit(counter), then index(k) -> position of this thread's item of sub-block k.
// All 256 threads call every method, with the same k.
constexpr int kAppendBatch = 8;
struct BatchAppend {
    uint32_t bits = 0;
    uint32_t (*cnt)[4];
    uint32_t *base;
    RT_DEV void note(int k, bool pred) {
        unsigned long long m = __ballot(pred);
        if (pred) bits |= 1u << k;
        if ((threadIdx.x & 63) == 0) cnt[k][threadIdx.x >> 6] = (uint32_t)__popcll(m);
    }
    RT_DEV void commit(uint32_t *counter) {
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t tot = 0;
            for (int k = 0; k < kAppendBatch; ++k) tot += cnt[k][0] + cnt[k][1] + cnt[k][2] + cnt[k][3];
            *base = tot ? atomicAdd(counter, tot) : 0u;
        }
        __syncthreads();
    }
    RT_DEV bool mine(int k) const { return (bits >> k) & 1u; }
    RT_DEV uint32_t index(int k) const {
        const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        uint32_t off = *base;
        for (int kk = 0; kk < k; ++kk) off += cnt[kk][0] + cnt[kk][1] + cnt[kk][2] + cnt[kk][3];
        for (uint32_t i = 0; i < wv; ++i) off += cnt[k][i];
        const unsigned long long m = __ballot(mine(k));
        return off + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    }
};
#define RT_BATCH_APPEND(name) __shared__ uint32_t name##_cnt[kAppendBatch][4]; __shared__ uint32_t name##_base; BatchAppend name; name.cnt = name##_cnt; name.base = &name##_base

// ---- stage: primary ----------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_primary(const DevFrame *__restrict__ fr, Targets tg, WaveBuf wb) {   // workgroup = kAppendBatch tiles
    const RtUniforms &u = fr->u;
    RT_BATCH_APPEND(ap);
    for (int k = 0; k < kAppendBatch; ++k) {
        const int tile = blockIdx.x * kAppendBatch + k;
        int px, py;
        const bool live = tile < fr->g.nLocalTiles * max(fr->g.batch, 1) && pixel_of_slot(fr->g, tile, threadIdx.x, px, py);
        const int slot = tile * 256 + threadIdx.x;
        bool cand = false;
        if (live) {
            V3 dir = primaryDirK(fr, sub_frame_of_tile(fr->g, tile), px, py);
            V3 rdInv = mk3(1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z);
            float tmin;
            cand = fr->sc.hasBVH && slab(ld3(u.camPos), rdInv, ld3(fr->sc.rootMin), ld3(fr->sc.rootMax), tmin) && !(tmin > u.inf);
            if (!cand) finish_miss(fr, wb, slot, px, py, dir);
        }
        ap.note(k, cand);
    }
    ap.commit(&wb.counts[0]);
    for (int k = 0; k < kAppendBatch; ++k) {
        const uint32_t idx = ap.index(k);
        if (ap.mine(k)) wb.cand[idx] = (uint32_t)((blockIdx.x * kAppendBatch + k) * 256 + threadIdx.x);
    }
}

// ---- persistent traversal ----------------------------------------------------------------------
// Ray sources.
struct PrimarySrc {   // ray i = primary ray of candidate i
    const DevFrame *fr;
    const uint32_t *cand;
    const uint32_t *count;
    float *outT;
    int *outTri;
    RT_DEV void prepare() {}
    RT_DEV uint32_t size() const { return *count; }
    // probe / take: see QueueSrc.  Every candidate is a ray; what the window read brings in is the candidate's pixel slot.
    struct Payload { uint32_t slot; };
    RT_DEV float probe(uint32_t i, Payload &p) const { p.slot = cand[i]; return fr->u.inf; }
    RT_DEV static Payload route(const Payload &p, int e) { Payload q; q.slot = (uint32_t)__shfl((int)p.slot, e, 64); return q; }
    RT_DEV void take(uint32_t i, const Payload &p, V3 &ro, V3 &rd, uint32_t &token) const {
        token = i;
        int px, py;
        slot_to_pixel(fr->g, p.slot, px, py);
        ro = ld3(fr->u.camPos);
        rd = primaryDirK(fr, sub_frame_of_slot(fr->g, p.slot), px, py);
    }
    RT_DEV void store_closest(uint32_t i, float t, int tri) const { outT[i] = t; outTri[i] = tri; }
    RT_DEV void store_any(uint32_t, bool) const {}
    RT_DEV bool dense(uint32_t, uint32_t) const { return false; }
    RT_DEV float probe_take(uint32_t, V3 &, V3 &, uint32_t &) const { return -1.0f; }
};
struct QueueSrc {     // slot-major queue: ray r -> (slot = r / n, j = r % n) at [slot*stride + j], n = live entries
    const float4 *o, *d;
    const float *tm;             // per-slot tMax / liveness
    const uint32_t *liveCount;   // device counter the live entry count derives from
    uint32_t c0, cap, stride, slots;
    uint32_t denseSlots;         // the first `denseSlots` slots hold a ray for (nearly) every entry (AO slots, the bounce queue): see dense() below
    float *outT;
    int *outTri;
    uint8_t *outOcc;
    uint32_t nLive;              // cached by prepare(): the count is final before this kernel starts
    RT_DEV void prepare() { uint32_t h = *liveCount; nLive = min(h, c0 + cap) - min(h, c0); }   // no wrapping subtraction, see chunk_live
    RT_DEV uint32_t size() const { return nLive * slots; }
    RT_DEV uint32_t addr(uint32_t r) const { return (r / nLive) * stride + (r % nLive); }
    // probe(r): window lane `lane` reads slot r's 4-byte liveness / tMax word (< 0 = no ray was cast into this slot); consecutive
    // r are consecutive words, so a 64-lane probe is one coalesced 256-byte read and dead slots (the disk-light samples of
    // surfaces facing away from the light, the sun / point rays of samples > 0) never touch their 32-byte records.  The scheduler
    // routes the queue address of each live slot to the lane that takes it (route: a cross-lane move); take: its record.
    // (Reading the records together with the liveness words -- one round trip per refill instead of two -- was measured slower for
    // the shadow queue, where 55 % of the slots are dead: 1.07 vs 1.01 ms.)
    struct Payload { uint32_t a; };
    RT_DEV float probe(uint32_t r, Payload &p) const { p.a = addr(r); return tm[p.a]; }
    RT_DEV static Payload route(const Payload &p, int e) { Payload q; q.a = (uint32_t)__shfl((int)p.a, e, 64); return q; }
    RT_DEV void take(uint32_t, const Payload &p, V3 &ro, V3 &rd, uint32_t &token) const {
        token = p.a;                     // results go to the same queue address: no second div/mod at retirement
        const float4 oo = o[p.a], dd = d[p.a];
        ro = f4xyz(oo); rd = f4xyz(dd);
    }
    RT_DEV void store_closest(uint32_t a, float t, int tri) const { outT[a] = t; outTri[a] = tri; }
    RT_DEV void store_any(uint32_t a, bool occ) const { outOcc[a] = occ ? 1 : 0; }
    // Dense slots (round 4): where (nearly) every entry is a ray the liveness probe is a wasted round trip -- the i-th idle lane takes the i-th entry
    // left and reads liveness word and record together; an entry that is dead after all (AO radius 0, GI switched off) just leaves its lane idle.
    RT_DEV bool dense(uint32_t r0, uint32_t r1) const { return r1 > r0 && (r1 - 1u) / nLive < denseSlots; }
    RT_DEV float probe_take(uint32_t r, V3 &ro, V3 &rd, uint32_t &token) const {
        const uint32_t a = addr(r);
        const float t = tm[a];
        const float4 oo = o[a], dd = d[a];
        token = a;
        ro = f4xyz(oo); rd = f4xyz(dd);
        return t;
    }
};

// Two any-hit queues traced by ONE persistent launch (direct shadows + AO, then the shadows at the bounce hits): a second
// launch would pay the ~0.15 ms ramp-up / drain latency of a persistent grid again for a few thousand rays.
struct DualQueueSrc {
    QueueSrc a, b;
    uint32_t na;
    RT_DEV void prepare() { a.prepare(); b.prepare(); na = a.size(); }
    RT_DEV uint32_t size() const { return na + b.size(); }
    typedef QueueSrc::Payload Payload;
    RT_DEV float probe(uint32_t r, Payload &p) const {
        if (r < na) return a.probe(r, p);
        const float t = b.probe(r - na, p);
        p.a |= 0x80000000u;               // results of the second queue (addresses stay below 2^31: checked on the host)
        return t;
    }
    RT_DEV static Payload route(const Payload &p, int e) { return QueueSrc::route(p, e); }
    RT_DEV void take(uint32_t r, const Payload &p, V3 &ro, V3 &rd, uint32_t &token) const {
        Payload q;
        q.a = p.a & 0x7fffffffu;
        if (p.a & 0x80000000u) b.take(r, q, ro, rd, token); else a.take(r, q, ro, rd, token);
        token = p.a;
    }
    RT_DEV void store_closest(uint32_t, float, int) const {}
    RT_DEV void store_any(uint32_t token, bool occ) const {
        if (token & 0x80000000u) b.outOcc[token & 0x7fffffffu] = occ ? 1 : 0;
        else a.outOcc[token] = occ ? 1 : 0;
    }
    RT_DEV bool dense(uint32_t r0, uint32_t r1) const { return r1 <= na && a.dense(r0, r1); }
    RT_DEV float probe_take(uint32_t r, V3 &ro, V3 &rd, uint32_t &token) const { return a.probe_take(r, ro, rd, token); }
};

// A dense list of queue addresses (rt_hybrid.hip): ray r is the record at idx[r]; every listed record is a ray.
struct IndexedSrc {
    const uint32_t *idx;
    const uint32_t *count;
    const float4 *o, *d;
    float *outT;
    int *outTri;
    uint32_t n;
    RT_DEV void prepare() { n = *count; }
    RT_DEV uint32_t size() const { return n; }
    struct Payload { uint32_t a; };
    RT_DEV float probe(uint32_t r, Payload &p) const { p.a = idx[r]; return 1.0f; }
    RT_DEV static Payload route(const Payload &p, int e) { Payload q; q.a = (uint32_t)__shfl((int)p.a, e, 64); return q; }
    RT_DEV void take(uint32_t, const Payload &p, V3 &ro, V3 &rd, uint32_t &token) const {
        token = p.a;
        const float4 oo = o[p.a], dd = d[p.a];
        ro = f4xyz(oo); rd = f4xyz(dd);
    }
    RT_DEV void store_closest(uint32_t a, float t, int tri) const { outT[a] = t; outTri[a] = tri; }
    RT_DEV void store_any(uint32_t, bool) const {}
    RT_DEV bool dense(uint32_t, uint32_t) const { return false; }
    RT_DEV float probe_take(uint32_t, V3 &, V3 &, uint32_t &) const { return -1.0f; }
};

// A dense array of ray records (rt_hybrid.hip, round 4): ray r is the record o[r] / d[r]; its answer goes to outT / outTri at dst[r] (the asking
// thread's log entry).  The list length is read on the device and clipped to the array's capacity (an overflowing pass is redone by the host).
struct CompactSrc {
    const float4 *o, *d;
    const uint32_t *dst;
    const uint32_t *count;
    const uint32_t *flags;   // bits 2 | 4: a pass outgrew its arrays -- the queue is incomplete and must not be traced
    uint32_t cap;
    uint32_t capOut;         // RT_HYBRID_CHECK=1: entries of outT / outTri; an answer addressed beyond them raises bit 32 of *flags instead of being stored (0: unchecked)
    float *outT;
    int *outTri;
    uint32_t n;
    RT_DEV void prepare() { n = (*flags & 6u) ? 0u : min(*count, cap); }
    RT_DEV uint32_t size() const { return n; }
    struct Payload { uint32_t a; };
    RT_DEV float probe(uint32_t r, Payload &p) const { p.a = r; return 1.0f; }
    RT_DEV static Payload route(const Payload &p, int e) { Payload q; q.a = (uint32_t)__shfl((int)p.a, e, 64); return q; }
    RT_DEV void take(uint32_t, const Payload &p, V3 &ro, V3 &rd, uint32_t &token) const {
        token = dst[p.a];
        const float4 oo = o[p.a], dd = d[p.a];
        ro = f4xyz(oo); rd = f4xyz(dd);
    }
    RT_DEV void store_closest(uint32_t a, float t, int tri) const {
        if (capOut && a >= capOut) { atomicOr(const_cast<uint32_t *>(flags), 32u); return; }
        outT[a] = t; outTri[a] = tri;
    }
    RT_DEV void store_any(uint32_t, bool) const {}
    RT_DEV bool dense(uint32_t, uint32_t) const { return false; }
    RT_DEV float probe_take(uint32_t, V3 &, V3 &, uint32_t &) const { return -1.0f; }
};

// hipcc sinks loads into the branches that first use them (e.g. a triangle's v0 behind the determinant test), which turns
// one gather round trip into two or three dependent ones.  pin() makes a loaded record "used" right after the loads were
// issued, so the whole group is in flight together.
RT_DEV void pin(float4 &v) { asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w)); }
// The traversal kernels pin whole 128-bit registers tuples: with four 32-bit constraints the register allocator is free to move the
// components apart from the tuple the load wrote, and did (19 v_mov per 4-wide node visit, about a tenth of the step's vector instructions --
// and these launches are bound by vector-instruction issue).
typedef float v4f __attribute__((ext_vector_type(4)));
RT_DEV void pin(v4f &v) { asm volatile("" : "+v"(v)); }

// ---- quad-cooperative record fetch (round 4; microbenchmark tools/gather2.hip) ------------------------------------
// The four lanes of a quad fetch ONE 64-byte record per instruction -- lane q piece q of the record quad-lane k stands on, k = 0..3 -- so a
// quad's load touches one line instead of four; a 4x4 transpose inside the quad (two butterfly stages of v_cndmask_b32_dpp: the select and the
// cross-lane read in one instruction, 32 per record set) then gives every lane the four pieces of ITS record.  Arithmetic and visit order are
// untouched.  All 64 lanes must be enabled where this runs (DPP reads of disabled lanes return the old destination).
template <int CTRL> RT_DEV int quad_bcast_i(int v) { return __builtin_amdgcn_mov_dpp(v, CTRL, 0xf, 0xf, true); }
// one dword of a butterfly stage: a = keepA ? x : partner(y), b = keepB ? y : partner(x).  `a` is written while x, y are still read
// (early clobber); `b` is written by the last instruction and may reuse an input's register, so a stage needs one spare register, not eight.
// (the s_nop 1 sits INSIDE the block: a DPP source written by a VALU instruction needs two wait states, the assembler does not insert them in inline asm, and
// between two asm statements the compiler may place a copy for an operand -- ADVICE r04.  The s_mov that follows is a third instruction in between.)
#define RT_QT_PAIR(PERM, X, Y, A, B)                                                                                               \
    asm volatile("s_nop 1\n\ts_mov_b64 vcc, %[ka]\n\tv_cndmask_b32_dpp %[a], %[y], %[x], vcc quad_perm:" PERM " row_mask:0xf bank_mask:0xf\n\t" \
                 "s_mov_b64 vcc, %[kb]\n\tv_cndmask_b32_dpp %[b], %[x], %[y], vcc quad_perm:" PERM " row_mask:0xf bank_mask:0xf"      \
                 : [a] "=&v"(A), [b] "=v"(B) : [x] "v"(X), [y] "v"(Y), [ka] "s"(keepA), [kb] "s"(keepB) : "vcc")
RT_DEV void quad_transpose(v4f &r0, v4f &r1, v4f &r2, v4f &r3) {
    // in: r[k] of quad-lane q = piece q of record k; out: r[j] of quad-lane q = piece j of record q
    v4f a0, a1, a2, a3;
    {
        const unsigned long long keepA = 0x5555555555555555ull, keepB = 0xAAAAAAAAAAAAAAAAull;   // lane bit 0 clear / set
        RT_QT_PAIR("[1,0,3,2]", r0.x, r1.x, a0.x, a1.x); RT_QT_PAIR("[1,0,3,2]", r0.y, r1.y, a0.y, a1.y);
        RT_QT_PAIR("[1,0,3,2]", r0.z, r1.z, a0.z, a1.z); RT_QT_PAIR("[1,0,3,2]", r0.w, r1.w, a0.w, a1.w);
        RT_QT_PAIR("[1,0,3,2]", r2.x, r3.x, a2.x, a3.x); RT_QT_PAIR("[1,0,3,2]", r2.y, r3.y, a2.y, a3.y);
        RT_QT_PAIR("[1,0,3,2]", r2.z, r3.z, a2.z, a3.z); RT_QT_PAIR("[1,0,3,2]", r2.w, r3.w, a2.w, a3.w);
    }
    {
        const unsigned long long keepA = 0x3333333333333333ull, keepB = 0xCCCCCCCCCCCCCCCCull;   // lane bit 1 clear / set
        RT_QT_PAIR("[2,3,0,1]", a0.x, a2.x, r0.x, r2.x); RT_QT_PAIR("[2,3,0,1]", a0.y, a2.y, r0.y, r2.y);
        RT_QT_PAIR("[2,3,0,1]", a0.z, a2.z, r0.z, r2.z); RT_QT_PAIR("[2,3,0,1]", a0.w, a2.w, r0.w, r2.w);
        RT_QT_PAIR("[2,3,0,1]", a1.x, a3.x, r1.x, r3.x); RT_QT_PAIR("[2,3,0,1]", a1.y, a3.y, r1.y, r3.y);
        RT_QT_PAIR("[2,3,0,1]", a1.z, a3.z, r1.z, r3.z); RT_QT_PAIR("[2,3,0,1]", a1.w, a3.w, r1.w, r3.w);
    }
}

// lane position of the n-th (0-based) set bit of m (n < popcount(m)): binary search over popcounts
RT_DEV uint32_t nth_set(unsigned long long m, uint32_t n) {
    uint32_t pos = 0;
#pragma unroll
    for (uint32_t w = 32u; w; w >>= 1) {
        const uint32_t c = (uint32_t)__popcll((m >> pos) & ((1ull << w) - 1ull));
        if (n >= c) { n -= c; pos += w; }
    }
    return pos;
}

// Diagnostic build only: how many cache accesses the lanes that call this together cost the vector L1 when each reads the record `key`.
// The texture addresser takes a wave's 16-byte lane-loads four adjacent lanes at a time and merges lanes of such a quad that read
// the same bytes (tools/gather.hip mode 6 under rocprofv3 --pmc: four adjacent lanes on one record = 0.25 accesses per lane-load);
// lanes further apart are not merged (the kernels' PMC access counts are 2-3 x their wave-wide distinct-record counts).
//   -> number of (quad, record) pairs = lanes whose key differs from every lower active lane of their quad.
RT_DEV uint32_t wave_distinct(uint32_t key) {   // distinct keys among the calling lanes
    unsigned long long m = __ballot(1);
    uint32_t n = 0;
    while (m) {
        const uint32_t k = (uint32_t)__shfl((int)key, __ffsll((long long)m) - 1, 64);
        m &= ~__ballot(key == k);
        n++;
    }
    return n;
}
RT_DEV uint32_t quad_distinct(uint32_t key) {
    const unsigned long long am = __ballot(1);
    const uint32_t lane = threadIdx.x & 63u, q0 = lane & ~3u;
    bool first = true;
#pragma unroll
    for (uint32_t j = 0; j < 3; ++j) {
        const uint32_t kj = (uint32_t)__shfl((int)key, (int)(q0 + j), 64);
        if (j < (lane & 3u) && ((am >> (q0 + j)) & 1ull) && kj == key) first = false;
    }
    return (uint32_t)__popcll(__ballot(first));
}

// Tunables of the scheduler (overridable per context through RT_REFILL_MIN / RT_MIN_SEARCH for experiments).
constexpr uint32_t kShards = 64, kShardStride = 32;   // cursor shards per trace launch, uint32 words between them (128 B)
constexpr uint32_t kHeadWords = kShards * kShardStride;

struct TraceTune { int refillMin; int minSearch; int chunk; int leafb; int skipTraversal; int quadRefill; int coop; int leafbClosest; int nearFirst; int reverse; int guided; int chunkMax; int denseTake; int qnodes; int fused; int impl; int timing; };   // skipTraversal: diagnostic (RT_DEBUG_SKIP_TRAVERSAL)
// fused (RT_FUSED=1, measured option of round 5, off): closest-hit launches walk the fused records (DevScene::wF) when rt_upload_bvh built them -- the reference's
// visiting order in half the dependent round trips (bounce rays: 20.6 -> 11.2 steps, primary 17.1 -> 9.7), bit-identical, and 3-4 % SLOWER in every mode (batched,
// frame by frame, one rank of eight): the same number of 16-byte lane-loads per ray, and that number -- not the length of the dependency chain -- is what these
// launches cost (profiles/r05_experiments.txt 1)
static TraceTune default_tune() {
    TraceTune t{32, 16, 0, 2, 0, 0, 0, 2, 0, 1, 0, 768, 1, -1, 0, 0, 0};
    if (const char *e = getenv("RT_TRACE_TIMING")) t.timing = atoi(e);
    // impl (RT_IMPLICIT=1, measured option of round 5, off): closest-hit launches walk 48-byte records WITHOUT child references (three loads per node visit instead of
    // four) when every leaf of the tree sits at one depth (rt_upload_bvh) -- bit-identical, 25 % fewer node loads, and no faster (bounce launch 0.615 -> 0.628 ms per
    // frame, primary 0.253 -> 0.259): together with `fused` the second half of the finding that neither the loads nor the dependent steps of these launches can be
    // removed for time while their vector instructions stay (profiles/r05_experiments.txt 2)
    if (const char *e = getenv("RT_IMPLICIT")) t.impl = atoi(e);
    if (const char *e = getenv("RT_FUSED")) t.fused = atoi(e);
    return t;
}

// The slab test of rt_bvh.glsl:124-134 in two halves, so that the per-axis values of two child boxes can be merged into their parent's (k_trace, FUSE):
// slab_parts + slab_eval are slab() operation for operation.
struct SlabP { float sx, sy, sz, bx, by, bz; };
RT_DEV SlabP slab_parts(V3 ro, V3 rdInv, V3 bmin, V3 bmax) {
    const V3 t0 = (bmin - ro) * rdInv, t1 = (bmax - ro) * rdInv;
    SlabP p;
    p.sx = fminr(t0.x, t1.x); p.sy = fminr(t0.y, t1.y); p.sz = fminr(t0.z, t1.z);
    p.bx = fmaxr(t0.x, t1.x); p.by = fmaxr(t0.y, t1.y); p.bz = fmaxr(t0.z, t1.z);
    return p;
}
RT_DEV bool slab_eval(const SlabP &p, float &tminOut) {
    const float tmin = fmaxr(fmaxr(p.sx, p.sy), fmaxr(p.sz, 0.0f));
    const float tmax = fminr(fminr(p.bx, p.by), p.bz);
    tminOut = tmin;
    return tmax >= tmin;
}
// the parts of the box min(a.min, b.min) .. max(a.max, b.max): (x - ro) * rdInv is monotone in x, so the union's near / far plane distances are the smaller / larger
// of the two boxes' (v_min / v_max drop the NaN of 0 * inf and of an absent child's NaN box exactly as the direct evaluation does: tests/test_fused_nodes.py)
RT_DEV SlabP slab_union(const SlabP &a, const SlabP &b) {
    SlabP p;
    p.sx = fminr(a.sx, b.sx); p.sy = fminr(a.sy, b.sy); p.sz = fminr(a.sz, b.sz);
    p.bx = fmaxr(a.bx, b.bx); p.by = fmaxr(a.by, b.by); p.bz = fmaxr(a.bz, b.bz);
    return p;
}

template <bool ANY> struct StackOf { typedef StackEntry type; };          // closest: {deferred child, its entry distance}
template <> struct StackOf<true> { typedef uint32_t type; };              // any-hit: the pop-time cull never fires (tMax is constant)

// Per-lane traversal stacks live in dynamic LDS sized for THIS tree (stackEntries per lane): the resident workgroups per CU -- and
// with them the memory-level parallelism of these latency-bound loops -- follow from the scene's depth instead of from a few
// compiled-in sizes (1 M triangles, depth 18: 4 / 5 workgroups per CU for closest- / any-hit instead of 3 / 4 with 24- and 36-entry stacks).
extern __shared__ __align__(16) unsigned char rt_dyn_lds[];
// Register budget: the closest-hit launches run five workgroups per CU (their 8-byte stack entries fill the LDS first), so their kernels may use
// up to 96 VGPRs (five waves per SIMD) but not more; the diagnostic builds are unconstrained.
#ifndef RT_ANYHIT_WAVES
#define RT_ANYHIT_WAVES 7   // any-hit launches: 72 VGPRs, seven waves per SIMD (with the exact stack size of rt_upload_bvh seven workgroups fit a CU's LDS)
#endif
#ifndef RT_IMPL_ANYHIT_WAVES
#define RT_IMPL_ANYHIT_WAVES 6   // the implicit any-hit build at six waves per SIMD: at seven (72 VGPRs) it spills 32 B per lane into its inner loop (experiment 9)
#endif
template <class Src, bool ANY, int LEAFB, bool STATS = false, bool COOP = false, bool NEAR = false, int QN = 0, bool FUSE = false, bool IMPL = false, bool TIMING = false>
__global__ __launch_bounds__(256, (!STATS && !ANY) ? 5 : ((NEAR || QN == 2 || (IMPL && RT_IMPL_ANYHIT_WAVES == 6)) ? 6 : (ANY && !STATS && LEAFB == 2 ? RT_ANYHIT_WAVES : 1))) void k_trace(const DevFrame *__restrict__ fr, const float4 *__restrict__ wnodes, const float4 *__restrict__ tris, Src src,
                                                uint32_t *head, unsigned long long *tally, unsigned long long *gatherLoads, TraceTune tune,
                                                int stackEntries, unsigned long long *stats = nullptr, const float4 *__restrict__ leafBox = nullptr) {
    // STATS (diagnostic build only, RT_TRACE_STATS=1): [0] inner-node visits [1] leaf visits [2] triangle tests [3] inner-phase wave
    // iterations [4] active lanes summed over them [5] leaf-phase wave iterations [6] lanes with a leaf summed [7] refill rounds
    unsigned long long st_[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};   // [14] / [15]: the same with wave-wide merging (distinct records per wave step)   // [8] cycles in inner steps [9] in leaf phases [10] in refills [11] wave lifetime
    // [12] / [13] node / triangle gather loads after merging: adjacent lanes (a quad) that stand on the same record read the same 16-byte
    // pieces, which the vector L1 serves as one access (quad_distinct above) -- (quad, record) pairs per wave step x loads per record
    const unsigned long long tStart_ = (STATS || TIMING) ? clock64() : 0ull;
    // TIMING (RT_TRACE_TIMING=1, diagnostic build of the PRODUCTION kernels -- same registers and occupancy, nothing of the STATS build's counting): where a wave's cycles
    // go, from s_memtime stamps kept in scalar registers: [0] inner-phase iterations [1] cycles from the top of an iteration to the issue of its loads [2] from there to
    // their arrival (an explicit s_waitcnt vmcnt(0)) [3] from there to the end of the iteration [4] leaf phases [5] cycles in them [6] refill rounds [7] cycles in them
    // [8] wave lifetime
    unsigned long long tm_[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tA_ = 0, tB_ = 0, tC_ = 0;
    auto stamp_loads = [&]() {
        if constexpr (TIMING) { tB_ = clock64(); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); tC_ = clock64(); }
    };
    typedef typename StackOf<ANY>::type Entry;
    Entry *stk = reinterpret_cast<Entry *>(rt_dyn_lds) + (threadIdx.x >> 6) * stackEntries * 64 + (threadIdx.x & 63);
    DevScene sc = fr->sc;   // private copy: scene constants stay in SGPRs instead of being re-read per step
    sc.tris = tris;         // kernel-argument copies: known-global pointers (global_load, not flat_load)
    const float4 *__restrict__ nodes = wnodes;   // 2-wide records for closest-hit, 4-wide records for any-hit
    const float eps = fr->u.eps, inf = fr->u.inf;
    src.prepare();
    const uint32_t n = src.size();
    const uint32_t lane = threadIdx.x & 63;
    // run length: RT_CHUNK, else about a third of a wave's share of the queue, in [128, 384] rays (measured on MI355X, 1080p / 4 spp:
    // whole frame 2.42 / 2.26 / 2.18 / 2.20 ms with runs of 64 / 128 / 256 / 512; one rank of eight 0.51 / 0.48 / 0.52 / 0.55; round 3, batches
    // of eight frames, 1.83 / 1.78 / 1.79 ms per frame with runs of at most 256 / 384 / 512)
    // round 4 (seven-wave any-hit kernels, larger grids): at most 768 -- 1.669 against 1.678 / 1.695 / 1.70 ms per frame with runs of at most 512 / 384 / 1024
    // primary launch (chunkMax 256): runs of 64 candidates = one 8x8 pixel block were right for single frames traced alone (rounds 1-3); with batches of
    // frames and four launch sets in flight 256 are: 1.613 against 1.640 ms per frame, frame by frame 1.805 against 1.825, one rank of eight 0.225 against 0.230
    const uint32_t runLen = tune.chunk > 0 ? (uint32_t)max(tune.chunk, 8)
                                           : min((uint32_t)tune.chunkMax, max(128u, ((n / (3u * 4u * gridDim.x) + 63u) / 64u) * 64u));

    // per-lane ray state
    V3 ro = mk3(0.0f), rd = mk3(0.0f), rdInv = mk3(0.0f);
    float tBest = 0.0f;          // closest: best t so far; any: tMax
    int triBest = -1;
    int ref = 0, sp = 0;
    int leaf = 0;                // any-hit only: one postponed leaf (0 = none; leaf refs are negative)
    uint32_t rayId = 0;
    bool active = false;
    bool exhausted = (n == 0);
    uint32_t traced = 0;
    uint32_t gathers = 0;        // 16-byte gather loads this lane issued for nodes and triangles (the L1 gather roofline's unit)
    uint32_t runNext = 0, runEnd = 0;   // wave-uniform: the part of the current run not handed out yet
    const uint32_t shard = (blockIdx.x * 4u + (threadIdx.x >> 6)) % kShards;   // home shard of this wave
    bool homeDry = false;
    // Guided run lengths (RT_GUIDED=1 / 2, measured option of round 4, off): the rays dealt LAST go out in short runs -- `lateRays` (about an eighth of the
    // queue, a multiple of runLen) in runs of runLen / 8 (at least 64) -- so that the launch does not end with a few waves still working through a long run
    // each.  Alone the bounce launch gains 2 %; with four launch sets in flight, which fill each other's tails anyway, the frame is 1-2 % slower
    // (profiles/r04_experiments.txt 13).  Dealing order d = 0 .. nRuns - 1: the nEarly long runs first, then the short ones.
    const bool guided = tune.guided == 1 || (tune.guided == 2 && !ANY);   // 2: closest-hit launches only
    const uint32_t shortLen = guided ? max(64u, runLen / 8u) : runLen;
    const uint32_t lateRays = (guided && n > 16u * runLen) ? (n / 8u / runLen) * runLen : 0u;
    const uint32_t nEarly = (n - lateRays + runLen - 1u) / runLen;
    const uint32_t nRuns = nEarly + (lateRays + shortLen - 1u) / shortLen;

    // pop the next subtree of this lane's ray, or retire the ray
    auto pop_or_finish = [&]() {
        bool found = false;
        while (sp > 0) {
            sp--;
            if constexpr (ANY) {
                const int r = (int)stk[sp * 64];
                if (r < 0 && leaf == 0) { leaf = r; continue; }   // leaves wait in `leaf`; keep looking for an inner node
                ref = r;                                          // an inner node, or a second leaf (the lane then waits)
                found = true;
                break;
            } else {
                StackEntry e = stk[sp * 64];
                if (u2f(e.y) > tBest) continue;   // rt_bvh.glsl:208 cull
                ref = (int)e.x;
                found = true;
                break;
            }
        }
        if (!found) {
            if constexpr (ANY) {
                ref = RT_NO_CHILD;
                if (leaf == 0) { src.store_any(rayId, false); active = false; }   // else: the postponed leaf is all that is left
            } else {
                src.store_closest(rayId, triBest >= 0 ? tBest : inf, triBest);
                active = false;
            }
        }
    };

    for (;;) {
        // ---- scheduler: idle lanes take the next rays of this wave's current run (`runLen` consecutive rays: same ray
        // type, neighbouring pixels).  Runs are dealt by kShards cursors, each on its own 128-byte line: shard s owns
        // runs s, s+kShards, ...; a wave draws from its home shard and steals round-robin once that is dry.  (One shared
        // cursor word sustains only ~88 M atomics/s on MI355X -- per-refill, then per-run atomics on a single word
        // bounded earlier versions of this kernel; a purely static deal leaves the bounce-ray tail unbalanced.)
        // quadRefill: only quads of four adjacent lanes that are idle together take new rays, four consecutive ones: the vector L1 merges
        // the loads of adjacent lanes that stand on the same record (quad_distinct), and rays dealt together walk the top of the tree together
        auto whole_quads = [&](unsigned long long m) {
            if (!tune.quadRefill) return m;
            unsigned long long q = m & (m >> 1) & (m >> 2) & (m >> 3) & 0x1111111111111111ull;
            return q | (q << 1) | (q << 2) | (q << 3);
        };
        unsigned long long idleMask = whole_quads(__ballot(!active));
        int nIdle = __popcll(idleMask);
        if (!exhausted && nIdle >= tune.refillMin) {
            const unsigned long long tR_ = (STATS || TIMING) ? clock64() : 0ull;
            if (STATS && lane == 0) st_[7]++;
            if (runNext >= runEnd) {
                // shard s owns runs s, s + kShards, ...: k = 0 .. runsOf(s)-1
                auto runsOf = [&](uint32_t sh) { return sh < nRuns ? (nRuns - sh + kShards - 1u) / kShards : 0u; };
                uint32_t k = 0, from = shard;
                bool got = false;
                if (!homeDry) {                       // live phase: one atomic per run on the home shard
                    if (lane == 0) k = atomicAdd(&head[shard * kShardStride], 1u);
                    k = __shfl(k, 0, 64);
                    got = k < runsOf(shard);
                    homeDry = !got;
                }
                while (!got) {                        // stealing / drain: ONE 64-lane probe of all cursors, then one atomic
                    const uint32_t cur = __hip_atomic_load(&head[lane * kShardStride], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    unsigned long long open = __ballot(cur < runsOf(lane));
                    if (open == 0ull) break;
                    // first open shard at or after the home shard (rotate so that waves spread over the open shards)
                    const unsigned long long rot = (open >> shard) | (shard ? (open << (64u - shard)) : 0ull);
                    from = (shard + (uint32_t)__ffsll((long long)rot) - 1u) % kShards;
                    if (lane == 0) k = atomicAdd(&head[from * kShardStride], 1u);
                    k = __shfl(k, 0, 64);
                    got = k < runsOf(from);           // lost a race for the last run of that shard: probe again
                }
                if (!got) { exhausted = true; continue; }
                // any-hit queues are dealt from their END (round 4): the expensive rays -- bounce-hit shadow rays, then the disk / sun / point slots -- sit behind
                // the cheap AO slots, and a launch that starts with its longest work ends with short runs instead of a tail of long ones
                const uint32_t dIdx = k * kShards + from;           // place in the dealing order
                const bool late = dIdx >= nEarly;
                const uint32_t len = late ? shortLen : runLen;
                const uint32_t span = late ? lateRays : n - lateRays;   // rays of the region this run belongs to
                const uint32_t off = (late ? dIdx - nEarly : dIdx) * len;   // offset inside the region, in dealing order
                const uint32_t cnt = min(len, span - off);
                if (ANY && tune.reverse) {                          // region layout: [late | early], each dealt from its end
                    const uint32_t top = late ? lateRays : n;
                    runEnd = top - off;
                    runNext = runEnd - cnt;
                } else {                                            // [early | late], each dealt from its beginning
                    runNext = (late ? n - lateRays : 0u) + off;
                    runEnd = runNext + cnt;
                }
            }
            // Deal the run's LIVE rays to the idle lanes, one 64-slot window per pass: every lane probes one slot's liveness word
            // (coalesced), the live slots go to the idle lanes in order and dead slots cost nothing further.  (Before, a refill
            // handed out `idle lanes` consecutive slots dead or alive and went round again: with 55 % of the shadow queue's slots
            // dead, three rounds -- each a full scheduler iteration -- to fill half a wave.)
            // a lane starts on the ray it was given (record in ro / rd, any-hit limit tMax)
            auto start_ray = [&](float tMax, uint32_t token) {
                rayId = token;
                traced++;
                rdInv = mk3(1.0f / rd.x, 1.0f / rd.y, 1.0f / rd.z);
                tBest = ANY ? tMax : inf;
                triBest = -1;
                sp = 0;
                leaf = 0;
                ref = ANY ? sc.rootRef4 : sc.rootRefW;
                if (ANY && ref < 0) { leaf = ref; ref = RT_NO_CHILD; }   // single-leaf tree
                float tmin;
                bool in = sc.hasBVH && !tune.skipTraversal && slab(ro, rdInv, ld3(sc.rootMin), ld3(sc.rootMax), tmin) && !(tmin > tBest);
                if (in) active = true;
                else if (ANY) src.store_any(token, false);
                else src.store_closest(token, inf, -1);
            };
            unsigned long long idleLeft = idleMask;
            while (idleLeft != 0ull && runNext < runEnd) {
                const uint32_t window = min(64u, runEnd - runNext);
                if (tune.denseTake && src.dense(runNext, runNext + window)) {
                    // dense slots: no probe -- the i-th idle lane reads the i-th entry left, liveness word and record in ONE round trip
                    const uint32_t nIdleL = (uint32_t)__popcll(idleLeft);
                    const uint32_t nTake = min(window, nIdleL);
                    const uint32_t rank = (uint32_t)__popcll(idleLeft & ((1ull << lane) - 1ull));
                    if (((idleLeft >> lane) & 1ull) && rank < nTake) {
                        uint32_t token;
                        const float tMax = src.probe_take(runNext + rank, ro, rd, token);
                        if (!(tMax < 0.0f)) start_ray(tMax, token);
                    }
                    runNext += nTake;
                    idleLeft = whole_quads(__ballot(!active));
                    continue;
                }
                float t = -1.0f;
                typename Src::Payload pl{};
                if (lane < window) t = src.probe(runNext + lane, pl);
                const unsigned long long liveMask = __ballot(lane < window && !(t < 0.0f));
                const uint32_t nLiveW = (uint32_t)__popcll(liveMask), nIdleL = (uint32_t)__popcll(idleLeft);
                const uint32_t nTake = min(nLiveW, nIdleL);
                const uint32_t rank = (uint32_t)__popcll(idleLeft & ((1ull << lane) - 1ull));
                const bool takes = ((idleLeft >> lane) & 1ull) && rank < nTake;
                const uint32_t e = takes ? nth_set(liveMask, rank) : lane;      // window entry this lane takes
                const float tMax = __shfl(t, (int)e, 64);
                const typename Src::Payload mine = Src::route(pl, (int)e);
                if (takes) {
                    uint32_t token;
                    src.take(runNext + e, mine, ro, rd, token);
                    start_ray(tMax, token);
                }
                // all live slots taken: the window is used up; else everything before the first live slot left over
                runNext += nLiveW <= nIdleL ? window : nth_set(liveMask, nTake);
                idleLeft = whole_quads(__ballot(!active));                     // root misses may draw again
            }
            if (STATS && lane == 0) st_[10] += clock64() - tR_;
            if constexpr (TIMING) { tm_[6]++; tm_[7] += clock64() - tR_; }
            continue;   // lanes that drew a dead slot or a root miss may draw again
        }
        if (__ballot(active) == 0ull) {
            if (exhausted) break;
            continue;
        }
        // ---- phase 1: walk inner nodes until (almost) every live lane holds a leaf
        for (;;) {
            const bool searching = active && (uint32_t)ref < (uint32_t)RT_NO_CHILD;   // an inner node in hand
            const unsigned long long sm = __ballot(searching);
            if (sm == 0ull) break;
            if (__popcll(sm) < tune.minSearch && __ballot(active && (ANY ? leaf != 0 : ref < 0)) != 0ull) break;   // keep the leaf phase dense
            const unsigned long long tI_ = STATS ? clock64() : 0ull;
            if constexpr (TIMING) { tA_ = clock64(); tB_ = tC_ = 0; }
            if (STATS && lane == 0) { st_[3]++; st_[4] += (unsigned long long)__popcll(sm); }
            // COOP (closest-hit, RT_COOP=1): the quad fetches its searching lanes' records together, then transposes (see quad_transpose)
            v4f ca = {0, 0, 0, 0}, cb = ca, cc = ca, cd = ca;
            if constexpr (COOP && !ANY) {
                const uint32_t q = lane & 3u;
                const int me = searching ? ref : -1;
                const int i0 = quad_bcast_i<0x00>(me), i1 = quad_bcast_i<0x55>(me), i2 = quad_bcast_i<0xAA>(me), i3 = quad_bcast_i<0xFF>(me);
                const v4f *nv = reinterpret_cast<const v4f *>(nodes) + q;
                if (i0 >= 0) ca = nv[(size_t)i0 * 4];
                if (i1 >= 0) cb = nv[(size_t)i1 * 4];
                if (i2 >= 0) cc = nv[(size_t)i2 * 4];
                if (i3 >= 0) cd = nv[(size_t)i3 * 4];
                pin(ca); pin(cb); pin(cc); pin(cd);
                quad_transpose(ca, cb, cc, cd);
            }
            if (searching) {
                if (STATS) { st_[0]++; const uint32_t dk = quad_distinct((uint32_t)ref), dw = wave_distinct((uint32_t)ref); if (lane == (uint32_t)(__ffsll((long long)sm) - 1)) { st_[12] += dk * (ANY ? 7u : (IMPL ? 3u : 4u)); st_[14] += dw * (ANY ? 7u : (IMPL ? 3u : 4u)); } }
                gathers += ANY ? (QN ? 4u : 7u) : 4u;
                if constexpr (ANY) {
                    int r0, r1, r2, r3;
                    float t0, t1, t2, t3;
                    bool h0, h1, h2, h3;
                    // IMPL (RT_IMPLICIT=1, trees whose leaves sit at one depth D): `ref` = depth << 24 | path of an even-depth node; its children are (d + 2, 4p + j) -- leaves
                    // -(4p + j + 1) when d + 2 == D, the two leaves -(2p + j + 1) when d + 1 == D -- so the record carries no references: six loads instead of seven
                    // (quantised: three instead of four), at the node's pre-order position
                    uint32_t dN = 0, pN = 0, atN = 0;
                    auto impl_refs = [&]() {
                        const uint32_t D = (uint32_t)sc.implD;
                        if (dN + 1u == D) { r0 = -(int)(2u * pN + 1u); r1 = -(int)(2u * pN + 2u); r2 = RT_NO_CHILD; r3 = RT_NO_CHILD; }
                        else if (dN + 2u == D) { r0 = -(int)(4u * pN + 1u); r1 = -(int)(4u * pN + 2u); r2 = -(int)(4u * pN + 3u); r3 = -(int)(4u * pN + 4u); }
                        else { const int b = (int)(((dN + 2u) << 24) | (4u * pN)); r0 = b; r1 = b + 1; r2 = b + 2; r3 = b + 3; }
                    };
                    if constexpr (IMPL) {
                        dN = (uint32_t)ref >> 24; pN = (uint32_t)ref & 0x00ffffffu;
                        atN = dN - (uint32_t)__popc(pN) + (pN << ((uint32_t)sc.implD - dN));
                        gathers -= 1u;      // (one load fewer than counted above)
                    }
                    if constexpr (QN) {
                        // RT_QNODES: 64-byte node, child boxes as bytes on the node's own grid (rt_upload_bvh): box = fmaf(byte, 2^e, origin) per component,
                        // checked at upload to contain the child's box; then the slab test of the exact kernel on the decoded floats
                        const v4f *ndv = IMPL ? reinterpret_cast<const v4f *>(nodes) + (size_t)atN * 3 : reinterpret_cast<const v4f *>(nodes + (size_t)ref * 4);
                        v4f p0 = ndv[0], p1 = ndv[1], p2 = ndv[2], p3 = {0, 0, 0, 0};
                        if constexpr (!IMPL) p3 = ndv[3];
                        pin(p0); pin(p1); pin(p2); pin(p3);
                        if constexpr (IMPL) impl_refs();
                        else { r0 = (int)f2u(p3.x); r1 = (int)f2u(p3.y); r2 = (int)f2u(p3.z); r3 = (int)f2u(p3.w); }
                        const uint32_t ex = f2u(p0.w);
                        const float sx = u2f((ex & 0xffu) << 23), sy = u2f(((ex >> 8) & 0xffu) << 23), sz = u2f(((ex >> 16) & 0xffu) << 23);
                        const uint32_t lx = f2u(p1.x), ly = f2u(p1.y), lz = f2u(p1.z), hx = f2u(p1.w), hy = f2u(p2.x), hz = f2u(p2.y);
                        auto box = [&](int k, float &t) {
                            const V3 lo = mk3(__builtin_fmaf((float)((lx >> (8 * k)) & 0xffu), sx, p0.x), __builtin_fmaf((float)((ly >> (8 * k)) & 0xffu), sy, p0.y),
                                              __builtin_fmaf((float)((lz >> (8 * k)) & 0xffu), sz, p0.z));
                            const V3 hi = mk3(__builtin_fmaf((float)((hx >> (8 * k)) & 0xffu), sx, p0.x), __builtin_fmaf((float)((hy >> (8 * k)) & 0xffu), sy, p0.y),
                                              __builtin_fmaf((float)((hz >> (8 * k)) & 0xffu), sz, p0.z));
                            return slab(ro, rdInv, lo, hi, t) && t <= tBest;
                        };
                        h0 = box(0, t0) && r0 != RT_NO_CHILD;
                        h1 = box(1, t1) && r1 != RT_NO_CHILD;
                        h2 = box(2, t2) && r2 != RT_NO_CHILD;
                        h3 = box(3, t3) && r3 != RT_NO_CHILD;
                    } else {
                    // 4-wide node: up to four grandchild boxes per 128-byte record, order irrelevant for any-hit
                    // component-wise: [min.x x4][min.y x4][min.z x4][max.x x4][max.y x4][max.z x4][ref x4] = 7 loads, 8th piece unused
                    const v4f *ndv = IMPL ? reinterpret_cast<const v4f *>(nodes) + (size_t)atN * 6 : reinterpret_cast<const v4f *>(nodes + (size_t)ref * 8);
                    v4f q0 = ndv[0], q1 = ndv[1], q2 = ndv[2], q3 = ndv[3], q4 = ndv[4], q5 = ndv[5], q6 = {0, 0, 0, 0};
                    if constexpr (!IMPL) q6 = ndv[6];
                    pin(q0); pin(q1); pin(q2); pin(q3); pin(q4); pin(q5); pin(q6);
                    stamp_loads();
                    if constexpr (IMPL) impl_refs();
                    else { r0 = (int)f2u(q6.x); r1 = (int)f2u(q6.y); r2 = (int)f2u(q6.z); r3 = (int)f2u(q6.w); }
                    h0 = slab(ro, rdInv, mk3(q0.x, q1.x, q2.x), mk3(q3.x, q4.x, q5.x), t0) && t0 <= tBest;
                    h1 = slab(ro, rdInv, mk3(q0.y, q1.y, q2.y), mk3(q3.y, q4.y, q5.y), t1) && t1 <= tBest;
                    // absent children carry NaN boxes: with v_min/v_max NaN semantics their slab test is false, so all
                    // loads are issued up front and the four tests are branch-free (no dependent "is there a child" round trip)
                    h2 = slab(ro, rdInv, mk3(q0.z, q1.z, q2.z), mk3(q3.z, q4.z, q5.z), t2) && t2 <= tBest;
                    h3 = slab(ro, rdInv, mk3(q0.w, q1.w, q2.w), mk3(q3.w, q4.w, q5.w), t3) && t3 <= tBest;
                    }
                    // Any-hit order is free, so leaves are postponed: the first leaf met goes to `leaf`, the lane goes on with an
                    // inner child (or pops one), and leaves are tested in the leaf phase when (nearly) every lane holds one --
                    // both phases run with more lanes busy than when a lane stops at its first leaf.
                    int nxt = RT_NO_CHILD;
                    auto take = [&](bool h, int r) {
                        if (!h) return;
                        if (r < 0 && leaf == 0) { leaf = r; return; }
                        if (nxt == RT_NO_CHILD) { nxt = r; return; }
                        if (r >= 0 && nxt < 0) { const int t = nxt; nxt = r; r = t; }   // go on with the inner node, defer the leaf
                        stk[sp * 64] = (uint32_t)r;
                        sp++;
                    };
                    if constexpr (NEAR) {
                        // RT_NEAR_FIRST=1 (experiment, round 4): go on with the NEAREST inner child that was hit instead of the first in record order -- an
                        // occluded ray then tends to meet its occluder sooner; the answer (an OR over leaves) does not depend on the order
                        float kb = h0 && r0 >= 0 ? t0 : 3.0e38f;
                        int ib = 0;
                        const float k1 = h1 && r1 >= 0 ? t1 : 3.0e38f, k2 = h2 && r2 >= 0 ? t2 : 3.0e38f, k3 = h3 && r3 >= 0 ? t3 : 3.0e38f;
                        if (k1 < kb) { kb = k1; ib = 1; }
                        if (k2 < kb) { kb = k2; ib = 2; }
                        if (k3 < kb) { kb = k3; ib = 3; }
                        if (ib == 1) { const bool th = h0; h0 = h1; h1 = th; const int tr_ = r0; r0 = r1; r1 = tr_; }
                        if (ib == 2) { const bool th = h0; h0 = h2; h2 = th; const int tr_ = r0; r0 = r2; r2 = tr_; }
                        if (ib == 3) { const bool th = h0; h0 = h3; h3 = th; const int tr_ = r0; r0 = r3; r3 = tr_; }
                        // (a leaf child in position 0 goes to `leaf` or the stack as before: `take` prefers inner nodes for nxt)
                    }
                    take(h0, r0); take(h1, r1); take(h2, r2); take(h3, r3);
                    if (nxt == RT_NO_CHILD) pop_or_finish();
                    else ref = nxt;
                } else if constexpr (IMPL) {
                    // Implicit records (round 5, rt_upload_bvh): every leaf at depth D, so a node is (depth d, path p), `ref` = d << 24 | p, its children (d + 1, 2p)
                    // and (d + 1, 2p + 1), a leaf the reference -(p + 1); the record -- the two child boxes, 48 bytes, THREE loads instead of four -- sits at the
                    // node's pre-order position.  Boxes, tests, order and stack contents are those of the 64-byte records below.
                    const uint32_t dN = (uint32_t)ref >> 24, pN = (uint32_t)ref & 0x00ffffffu;
                    const uint32_t at = dN - (uint32_t)__popc(pN) + (pN << ((uint32_t)sc.implD - dN));
                    const v4f *ndv = reinterpret_cast<const v4f *>(nodes) + (size_t)at * 3;
                    v4f a = ndv[0], b = ndv[1], c = ndv[2];
                    pin(a); pin(b); pin(c);
                    stamp_loads();
                    gathers -= 1u;                                              // (4 were counted above)
                    float tL, tR;
                    const bool hitL = slab(ro, rdInv, mk3(a.x, a.y, a.z), mk3(a.w, b.x, b.y), tL) && tL <= tBest;
                    const bool hitR = slab(ro, rdInv, mk3(b.z, b.w, c.x), mk3(c.y, c.z, c.w), tR) && tR <= tBest;
                    const bool lastLevel = dN + 1u == (uint32_t)sc.implD;
                    const int refL = lastLevel ? -(int)(2u * pN + 1u) : (int)(((dN + 1u) << 24) | (2u * pN));
                    const int refR = lastLevel ? -(int)(2u * pN + 2u) : (int)(((dN + 1u) << 24) | (2u * pN + 1u));
                    if (hitL && hitR) {
                        const bool leftFirst = tL < tR;
                        StackEntry e;
                        e.x = (uint32_t)(leftFirst ? refR : refL);
                        e.y = f2u(leftFirst ? tR : tL);
                        stk[sp * 64] = e;
                        sp++;
                        ref = leftFirst ? refL : refR;
                    } else if (hitL || hitR) {
                        ref = hitL ? refL : refR;
                    } else pop_or_finish();
                } else if constexpr (FUSE) {
                    // Fused records (round 5, rt_upload_bvh): `ref` names a hub N -- an even-level inner node whose 128-byte record holds the 64-byte records of its
                    // two children A and B -- or, with kPair set, ONE of those children (a far child the step at N deferred).  A full step is the reference's step
                    // at N (rt_bvh.glsl:226-239: both child boxes, near one first, far one deferred with its entry distance) followed at once by its step at the
                    // near child X, from one round trip of eight loads; a pair step is the step at the deferred child alone (four loads).  The boxes of A and B are
                    // the unions of their children's (checked at upload), so their slab values are merged from the grandchildren's: same floats, same decisions,
                    // same visiting order as the two-wide walk below -- the stack holds what it would hold (one entry per binary level).
                    constexpr uint32_t kPair = 0x40000000u, kHalfB = 0x20000000u, kIdx = 0x1fffffffu;
                    const uint32_t code = (uint32_t)ref;
                    const bool pairStep = (code & kPair) != 0u;
                    const uint32_t hub = code & kIdx;
                    const v4f *ndv = reinterpret_cast<const v4f *>(nodes + (size_t)hub * 8) + ((pairStep && (code & kHalfB)) ? 4 : 0);
                    v4f x0 = ndv[0], x1 = ndv[1], x2 = ndv[2], x3 = ndv[3];     // the half the walk goes into (full step: A, swapped below when B is nearer)
                    v4f y0 = {0, 0, 0, 0}, y1 = y0, y2 = y0, y3 = y0;
                    if (!pairStep) { y0 = ndv[4]; y1 = ndv[5]; y2 = ndv[6]; y3 = ndv[7]; }
                    pin(x0); pin(x1); pin(x2); pin(x3); pin(y0); pin(y1); pin(y2); pin(y3);
                    gathers += pairStep ? 0u : 4u;      // (4 were counted above)
                    SlabP pX1 = slab_parts(ro, rdInv, mk3(x0.x, x0.y, x0.z), mk3(x1.x, x1.y, x1.z));
                    SlabP pX2 = slab_parts(ro, rdInv, mk3(x2.x, x2.y, x2.z), mk3(x3.x, x3.y, x3.z));
                    int g1 = (int)f2u(x0.w), g2 = (int)f2u(x1.w);              // X's children (g2 == RT_NO_CHILD: X is a leaf, g1 its reference)
                    bool enter = true;
                    if (!pairStep) {
                        const SlabP pY1 = slab_parts(ro, rdInv, mk3(y0.x, y0.y, y0.z), mk3(y1.x, y1.y, y1.z));
                        const SlabP pY2 = slab_parts(ro, rdInv, mk3(y2.x, y2.y, y2.z), mk3(y3.x, y3.y, y3.z));
                        float tA, tB;
                        const bool hitA = slab_eval(slab_union(pX1, pX2), tA) && tA <= tBest;
                        const bool hitB = slab_eval(slab_union(pY1, pY2), tB) && tB <= tBest;
                        const bool goB = hitB && !(hitA && tA < tB);           // both hit: leftFirst = tA < tB (rt_bvh.glsl:232)
                        enter = hitA || hitB;
                        if (hitA && hitB) {                                     // defer the far child: a leaf as itself, an inner node as (hub, half)
                            const int f1 = goB ? g1 : (int)f2u(y0.w), f2 = goB ? g2 : (int)f2u(y1.w);
                            StackEntry e;
                            e.x = f2 == RT_NO_CHILD ? (uint32_t)f1 : (kPair | (goB ? 0u : kHalfB) | hub);
                            e.y = f2u(goB ? tA : tB);
                            stk[sp * 64] = e;
                            sp++;
                        }
                        if (goB) { pX1 = pY1; pX2 = pY2; g1 = (int)f2u(y0.w); g2 = (int)f2u(y1.w); }
                    }
                    if (!enter) pop_or_finish();
                    else if (g2 == RT_NO_CHILD) ref = g1;                       // the near child is a leaf: the leaf phase takes it
                    else {                                                      // the reference's step at the near child X
                        float t1, t2;
                        const bool h1 = slab_eval(pX1, t1) && t1 <= tBest;
                        const bool h2 = slab_eval(pX2, t2) && t2 <= tBest;
                        if (h1 && h2) {
                            const bool leftFirst = t1 < t2;
                            StackEntry e;
                            e.x = (uint32_t)(leftFirst ? g2 : g1);
                            e.y = f2u(leftFirst ? t2 : t1);
                            stk[sp * 64] = e;
                            sp++;
                            ref = leftFirst ? g1 : g2;
                        } else if (h1 || h2) {
                            ref = h1 ? g1 : g2;
                        } else pop_or_finish();
                    }
                } else {
                    const float4 *nd = nodes + (size_t)ref * 4;
                    const v4f *ndv = reinterpret_cast<const v4f *>(nd);
                    v4f a, b, c, d;
                    if constexpr (COOP) { a = ca; b = cb; c = cc; d = cd; }
                    else { a = ndv[0]; b = ndv[1]; c = ndv[2]; d = ndv[3]; pin(a); pin(b); pin(c); pin(d); stamp_loads(); }
                    float tL, tR;
                    bool hitL = slab(ro, rdInv, mk3(a.x, a.y, a.z), mk3(b.x, b.y, b.z), tL) && tL <= tBest;
                    bool hitR = slab(ro, rdInv, mk3(c.x, c.y, c.z), mk3(d.x, d.y, d.z), tR) && tR <= tBest;
                    int refL = (int)f2u(a.w), refR = (int)f2u(b.w);
                    if (hitL && hitR) {
                        bool leftFirst = tL < tR;
                        StackEntry e;
                        e.x = (uint32_t)(leftFirst ? refR : refL);
                        e.y = f2u(leftFirst ? tR : tL);
                        stk[sp * 64] = e;
                        sp++;
                        ref = leftFirst ? refL : refR;
                    } else if (hitL || hitR) {
                        ref = hitL ? refL : refR;
                    } else pop_or_finish();
                }
            }
            if (STATS) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); if (lane == 0) st_[8] += clock64() - tI_; }
            if constexpr (TIMING) { const unsigned long long tD = clock64(); tm_[0]++; if (tB_) { tm_[1] += tB_ - tA_; tm_[2] += tC_ - tB_; tm_[3] += tD - tC_; } }
        }
        // ---- phase 2: leaves
        const unsigned long long tL_ = (STATS || TIMING) ? clock64() : 0ull;
        const int leafNow = ANY ? leaf : ref;   // any-hit: the postponed leaf; closest: the leaf the walk stopped at
        if (STATS) { unsigned long long lm = __ballot(active && leafNow < 0); if (lane == 0 && lm) { st_[5]++; st_[6] += (unsigned long long)__popcll(lm); } }
        if (active && leafNow < 0) {
            if (STATS) st_[1]++;
            int v = -leafNow - 1;
            int first = v >> 3, count = (v & 7) + 1;   // first: pair record, count: triangles
            if constexpr (IMPL) { first = v * sc.implR; count = 2; }   // implicit records: leaf v owns the records from v * R on; its count arrives with the first of them (below)
            bool done = false;
            // RT_QNODES: the leaf's exact box (the test the 112-byte node makes in the parent) -- fetched together with the first triangle group, tested after it
            bool boxOK = true;
            v4f lb0 = {0, 0, 0, 0}, lb1 = lb0;
            if constexpr (QN) {
                const uint32_t at = IMPL ? (uint32_t)v : (sc.leafBoxMagic ? __umulhi((uint32_t)first, sc.leafBoxMagic) : (uint32_t)first);   // implicit records: the leaf's ordinal
                const v4f *lb = reinterpret_cast<const v4f *>(leafBox) + (size_t)at * 2;
                lb0 = lb[0]; lb1 = lb[1];
                gathers += 2u;
            }
            auto gate = [&]() {
                if constexpr (QN) {
                    pin(lb0); pin(lb1);
                    float tb;
                    boxOK = slab(ro, rdInv, mk3(lb0.x, lb0.y, lb0.z), mk3(lb0.w, lb1.x, lb1.y), tb) && tb <= tBest;
                }
            };
            // Two triangles of a leaf share one 80-byte record (5 gather loads instead of 6).  Records of a leaf are contiguous:
            // fetch LEAFB/2 of them at a time so that the gather round trips of one group overlap (the array is padded, so no
            // bounds branch); test in leaf order.
            // LEAFB == 4 (closest-hit launches, round 4): groups of FOUR triangles (two records, 10 loads in flight) while at least four are left,
            // then pairs, then the odd tail -- a 5-triangle leaf is two dependent round trips instead of three.  The closest-hit launches are held
            // to five workgroups per CU by their LDS stacks, so the 96 registers of this form cost them no occupancy (the any-hit launch would
            // drop from six to five waves per SIMD: it keeps LEAFB == 2).
            auto group = [&](auto npc, int i) {
                constexpr int NPG = decltype(npc)::value;
                const float4 *t = sc.tris + (size_t)(first + (i >> 1)) * 5;
                gathers += 5u * NPG;
                if (STATS) { const unsigned long long am = __ballot(1); const uint32_t dk = quad_distinct((uint32_t)(first + (i >> 1))), dw = wave_distinct((uint32_t)(first + (i >> 1))); if (lane == (uint32_t)(__ffsll((long long)am) - 1)) { st_[13] += dk * 5u * NPG; st_[15] += dw * 5u * NPG; } }
                const v4f *tv = reinterpret_cast<const v4f *>(t);
                v4f rec[NPG][5];
#pragma unroll
                for (int k = 0; k < NPG; ++k) { rec[k][0] = tv[k * 5 + 0]; rec[k][1] = tv[k * 5 + 1]; rec[k][2] = tv[k * 5 + 2]; rec[k][3] = tv[k * 5 + 3]; rec[k][4] = tv[k * 5 + 4]; }
#pragma unroll
                for (int k = 0; k < NPG; ++k) { pin(rec[k][0]); pin(rec[k][1]); pin(rec[k][2]); pin(rec[k][3]); pin(rec[k][4]); }
                if (QN && i == 0) gate();
#pragma unroll
                for (int k = 0; k < 2 * NPG; ++k) {
                    const v4f &r0 = rec[k >> 1][0], &r1 = rec[k >> 1][1], &r2 = rec[k >> 1][2], &r3 = rec[k >> 1][3], &r4 = rec[k >> 1][4];
                    const V3 v0 = (k & 1) ? mk3(r2.y, r2.z, r2.w) : mk3(r0.x, r0.y, r0.z);
                    const V3 e1 = (k & 1) ? mk3(r3.x, r3.y, r3.z) : mk3(r0.w, r1.x, r1.y);
                    const V3 e2 = (k & 1) ? mk3(r3.w, r4.x, r4.y) : mk3(r1.z, r1.w, r2.x);
                    float tt;
                    if (STATS && !done) st_[2]++;
                    if (!done && boxOK && tri_hit(ro, rd, v0, e1, e2, eps, tBest, tt)) {
                        if (ANY) done = true;
                        else { tBest = tt; triBest = (int)f2u(r4.z) + (k & 1); }
                    }
                }
            };
            int i = 0;
            if constexpr (IMPL) {
                // first record of the leaf: one or two triangles and, in its spare word, the leaf's triangle count
                const v4f *tv = reinterpret_cast<const v4f *>(sc.tris + (size_t)first * 5);
                gathers += 5u;
                v4f r0 = tv[0], r1 = tv[1], r2 = tv[2], r3 = tv[3], r4 = tv[4];
                pin(r0); pin(r1); pin(r2); pin(r3); pin(r4);
                count = (int)f2u(r4.w);
                if (STATS) st_[2] += (unsigned long long)min(count, 2);
                if (QN) gate();                            // quantised nodes: the leaf's exact box, fetched with this record, decides whether its triangles count
                float tt;
                if (boxOK && tri_hit(ro, rd, mk3(r0.x, r0.y, r0.z), mk3(r0.w, r1.x, r1.y), mk3(r1.z, r1.w, r2.x), eps, tBest, tt)) {
                    if (ANY) done = true;
                    else { tBest = tt; triBest = (int)f2u(r4.z); }
                }
                if (boxOK && !done && count >= 2 && tri_hit(ro, rd, mk3(r2.y, r2.z, r2.w), mk3(r3.x, r3.y, r3.z), mk3(r3.w, r4.x, r4.y), eps, tBest, tt)) {
                    if (ANY) done = true;
                    else { tBest = tt; triBest = (int)f2u(r4.z) + 1; }
                }
                i = 2;
                if (count == 1) count = 2;                 // (nothing left: keeps the odd-tail test below from firing for a one-triangle leaf)
            }
            if constexpr (LEAFB >= 4) for (; i + 4 <= count && !done; i += 4) group(std::integral_constant<int, 2>{}, i);
            for (; i + 2 <= count && !done && boxOK; i += 2) group(std::integral_constant<int, 1>{}, i);
            // ... then the odd last triangle of the leaf: its record holds ONE triangle, whose nine floats (and, for closest-hit rays, its
            // index, repeated in the otherwise unused tenth float) sit in the first three 16-byte pieces: 3 gather loads instead of 5
            // (every leaf of the bench mesh has 5 triangles: 13 loads per leaf visit instead of 15).
            if ((count & 1) && !done && boxOK) {
                const float4 *t = sc.tris + (size_t)(first + (count >> 1)) * 5;
                gathers += 3u;
                if (STATS) { const unsigned long long am = __ballot(1); const uint32_t dk = quad_distinct((uint32_t)(first + (count >> 1))), dw = wave_distinct((uint32_t)(first + (count >> 1))); if (lane == (uint32_t)(__ffsll((long long)am) - 1)) { st_[13] += dk * 3u; st_[15] += dw * 3u; } st_[2]++; }
                const v4f *tv = reinterpret_cast<const v4f *>(t);
                v4f r0 = tv[0], r1 = tv[1], r2 = tv[2];
                pin(r0); pin(r1); pin(r2);
                if (QN && count == 1) gate();
                float tt;
                if (boxOK && tri_hit(ro, rd, mk3(r0.x, r0.y, r0.z), mk3(r0.w, r1.x, r1.y), mk3(r1.z, r1.w, r2.x), eps, tBest, tt)) {
                    if (ANY) done = true;
                    else { tBest = tt; triBest = (int)f2u(r2.y); }
                }
            }
            if (ANY && done) {
                src.store_any(rayId, true);
                active = false;
            } else if constexpr (ANY) {
                leaf = 0;
                if (ref < 0) { leaf = ref; ref = RT_NO_CHILD; }        // a second leaf was waiting in `ref`
                if (ref == RT_NO_CHILD) pop_or_finish();               // look for an inner node (retires the ray if nothing is left)
            } else pop_or_finish();
        }
        if (STATS && lane == 0) st_[9] += clock64() - tL_;
        if constexpr (TIMING) { tm_[4]++; tm_[5] += clock64() - tL_; }
    }
    if (STATS && lane == 0) st_[11] = clock64() - tStart_;
    if constexpr (TIMING) { tm_[8] = clock64() - tStart_; if (stats && lane == 0) for (int q = 0; q < 9; ++q) atomicAdd(&stats[q], tm_[q]); }
    if (STATS && stats) {
        for (int q = 0; q < 16; ++q) {
            unsigned long long v = st_[q];
            for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
            if (lane == 0 && v) atomicAdd(&stats[q], v);
        }
    }
    if (tally) {
        unsigned long long s = traced;                     // 64-bit: a batch of 64-spp frames traces more than 2^32 rays per launch set
        for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
        if (lane == 0 && s) atomicAdd(tally, s);
    }
    if (gatherLoads) {
        unsigned long long g = gathers;
        for (int off = 32; off > 0; off >>= 1) g += __shfl_down(g, off, 64);
        if (lane == 0 && g) atomicAdd(gatherLoads, g);
    }
}

// ---- packet traversal (round 4) ----------------------------------------------------------------------
// The AO rays of one hit leave from ONE point (computeAO: org = hp + N * aoBias, rt_lighting.glsl:721-757) and are short (aoRadius 0.8): the
// four of them visit nearly the same part of the tree -- traced one by one they fetch 39.8 four-wide node records and 8.8 leaves per hit on the
// bench mesh, as a packet 14.0 and 3.7 (tools/r04_packet_proto.py).  They are 83 % of the any-hit rays of a frame, and the any-hit launch runs at
// the vector L1's access rate.  So one LANE walks the tree once for the up to four rays of a hit: every node record and every
// triangle record is fetched once and tested against the rays that entered its parent (a 4-bit mask travels with every stack entry), a ray leaves
// the packet when it is occluded.  Every (ray, box) and (ray, triangle) test is the arithmetic of the single-ray kernel on the same operands, and a
// ray's answer is the OR over the reference's leaves whose own box the ray passes ("any-hit rays walk 4-wide nodes") -- which is
// what this computes, in another order -- so the answers are the same bits.
struct PacketSrc {       // packet p -> hit j = p % nLive, ray group g = p / nLive: rays i = 4 g .. 4 g + 3 (< A) at [i * stride + j]
    const float4 *o, *d;
    const float *tm;
    uint8_t *occ;
    const uint32_t *liveCount;
    uint32_t c0, cap, stride;
    int A;
    uint32_t nLive;
    RT_DEV void prepare() { uint32_t h = *liveCount; nLive = min(h, c0 + cap) - min(h, c0); }
    RT_DEV uint32_t size() const { return nLive * (uint32_t)((A + 3) / 4); }
};
// stack / register entry: [31:28] rays of the packet that passed the node's box, [27] leaf, [26:0] inner node index or leaf code (first << 3 | count - 1); 0 = none
RT_DEV uint32_t pk_entry(uint32_t mask, int ref) { return (mask << 28) | (ref < 0 ? (0x08000000u | (uint32_t)(-ref - 1)) : (uint32_t)ref); }
RT_DEV bool pk_is_leaf(uint32_t e) { return (e & 0x08000000u) != 0u; }

__global__ __launch_bounds__(256, 4) void k_trace_packets(const DevFrame *__restrict__ fr, const float4 *__restrict__ w4, const float4 *__restrict__ pairs, PacketSrc src,
                                                          uint32_t *head, unsigned long long *tally, unsigned long long *gatherLoads, TraceTune tune, int stackEntries) {
    uint32_t *stk = reinterpret_cast<uint32_t *>(rt_dyn_lds) + (threadIdx.x >> 6) * stackEntries * 64 + (threadIdx.x & 63);
    const DevScene sc = fr->sc;
    const float eps = fr->u.eps;
    src.prepare();
    const uint32_t n = src.size();
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t runLen = tune.chunk > 0 ? (uint32_t)max(tune.chunk, 8) : min(256u, max(64u, ((n / (3u * 4u * gridDim.x) + 63u) / 64u) * 64u));
    // per-lane packet state
    V3 ro = mk3(0.0f), rd[4], rdInv[4];
    float tMax[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { rd[r] = mk3(0.0f); rdInv[r] = mk3(0.0f); tMax[r] = -1.0f; }
    uint32_t alive = 0;          // rays of the packet not decided yet
    uint32_t occl = 0;           // rays found occluded
    uint32_t cur = 0, leafE = 0; // node in hand / postponed leaf (entries, 0 = none)
    int sp = 0;
    uint32_t token = 0;          // address of the packet's first ray
    uint32_t nRays = 0;          // rays of the packet that exist (i < A)
    bool active = false;
    bool exhausted = (n == 0);
    uint32_t traced = 0, gathers = 0;
    uint32_t runNext = 0, runEnd = 0;
    const uint32_t shard = (blockIdx.x * 4u + (threadIdx.x >> 6)) % kShards;
    bool homeDry = false;
    const uint32_t nRuns = (n + runLen - 1u) / runLen;

    auto retire = [&]() {
#pragma unroll
        for (int r = 0; r < 4; ++r) if ((uint32_t)r < nRays && !(tMax[r] < 0.0f)) src.occ[token + (uint32_t)r * src.stride] = (occl >> r) & 1u;
        active = false;
    };
    auto pop_or_finish = [&]() {
        cur = 0;
        while (sp > 0) {
            sp--;
            uint32_t e = stk[sp * 64];
            const uint32_t m = (e >> 28) & alive;
            if (m == 0u) continue;                              // every ray that entered this subtree has been decided since
            e = (e & 0x0fffffffu) | (m << 28);
            if (pk_is_leaf(e) && leafE == 0u) { leafE = e; continue; }   // leaves wait; keep looking for an inner node
            cur = e;                                                      // an inner node, or a second leaf (the lane then waits)
            break;
        }
        if (cur == 0u && leafE == 0u) retire();
    };

    for (;;) {
        const unsigned long long idleMask = __ballot(!active);
        const int nIdle = __popcll(idleMask);
        if (!exhausted && nIdle >= tune.refillMin) {
            if (runNext >= runEnd) {
                auto runsOf = [&](uint32_t sh) { return sh < nRuns ? (nRuns - sh + kShards - 1u) / kShards : 0u; };
                uint32_t k = 0, from = shard;
                bool got = false;
                if (!homeDry) {
                    if (lane == 0) k = atomicAdd(&head[shard * kShardStride], 1u);
                    k = __shfl(k, 0, 64);
                    got = k < runsOf(shard);
                    homeDry = !got;
                }
                while (!got) {
                    const uint32_t c = __hip_atomic_load(&head[lane * kShardStride], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    unsigned long long open = __ballot(c < runsOf(lane));
                    if (open == 0ull) break;
                    const unsigned long long rot = (open >> shard) | (shard ? (open << (64u - shard)) : 0ull);
                    from = (shard + (uint32_t)__ffsll((long long)rot) - 1u) % kShards;
                    if (lane == 0) k = atomicAdd(&head[from * kShardStride], 1u);
                    k = __shfl(k, 0, 64);
                    got = k < runsOf(from);
                }
                if (!got) { exhausted = true; continue; }
                const unsigned long long base = ((unsigned long long)k * kShards + from) * runLen;
                runNext = (uint32_t)base;
                runEnd = (uint32_t)min((unsigned long long)n, base + runLen);
            }
            // every packet of the run exists: the i-th idle lane takes the i-th packet left
            const uint32_t nTake = min((uint32_t)nIdle, runEnd - runNext);
            const uint32_t rank = (uint32_t)__popcll(idleMask & ((1ull << lane) - 1ull));
            if (!active && rank < nTake) {
                const uint32_t p = runNext + rank;
                const uint32_t g = p / src.nLive, j = p % src.nLive;
                token = (4u * g) * src.stride + j;
                nRays = min(4u, (uint32_t)src.A - 4u * g);
                ro = f4xyz(src.o[token]);        // one origin for the packet
                alive = 0; occl = 0; sp = 0; cur = 0; leafE = 0;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    tMax[r] = -1.0f;
                    if ((uint32_t)r < nRays) {
                        const uint32_t a = token + (uint32_t)r * src.stride;
                        tMax[r] = src.tm[a];
                        if (!(tMax[r] < 0.0f)) {
                            rd[r] = f4xyz(src.d[a]);
                            rdInv[r] = mk3(1.0f / rd[r].x, 1.0f / rd[r].y, 1.0f / rd[r].z);
                            traced++;
                            float tmin;
                            if (sc.hasBVH && !tune.skipTraversal && slab(ro, rdInv[r], ld3(sc.rootMin), ld3(sc.rootMax), tmin) && !(tmin > tMax[r])) alive |= 1u << r;
                        }
                    }
                }
                active = true;
                if (alive == 0u) retire();
                else {
                    const uint32_t e = pk_entry(alive, sc.rootRef4);
                    if (pk_is_leaf(e)) leafE = e; else cur = e;   // (single-leaf tree)
                }
            }
            runNext += nTake;
            continue;
        }
        if (__ballot(active) == 0ull) {
            if (exhausted) break;
            continue;
        }
        // ---- inner nodes
        for (;;) {
            const bool searching = active && cur != 0u && !pk_is_leaf(cur);
            const unsigned long long sm = __ballot(searching);
            if (sm == 0ull) break;
            if (__popcll(sm) < tune.minSearch && __ballot(active && leafE != 0u) != 0ull) break;
            if (searching) {
                gathers += 7u;
                const uint32_t pm = cur >> 28;
                const v4f *ndv = reinterpret_cast<const v4f *>(w4 + (size_t)(cur & 0x07ffffffu) * 8);
                v4f q0 = ndv[0], q1 = ndv[1], q2 = ndv[2], q3 = ndv[3], q4 = ndv[4], q5 = ndv[5], q6 = ndv[6];
                pin(q0); pin(q1); pin(q2); pin(q3); pin(q4); pin(q5); pin(q6);
                uint32_t cm0 = 0, cm1 = 0, cm2 = 0, cm3 = 0;   // per child: the rays that pass its box
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float t0, t1, t2, t3;
                    const bool h0 = slab(ro, rdInv[r], mk3(q0.x, q1.x, q2.x), mk3(q3.x, q4.x, q5.x), t0) && t0 <= tMax[r];
                    const bool h1 = slab(ro, rdInv[r], mk3(q0.y, q1.y, q2.y), mk3(q3.y, q4.y, q5.y), t1) && t1 <= tMax[r];
                    const bool h2 = slab(ro, rdInv[r], mk3(q0.z, q1.z, q2.z), mk3(q3.z, q4.z, q5.z), t2) && t2 <= tMax[r];
                    const bool h3 = slab(ro, rdInv[r], mk3(q0.w, q1.w, q2.w), mk3(q3.w, q4.w, q5.w), t3) && t3 <= tMax[r];
                    cm0 |= (h0 ? 1u : 0u) << r; cm1 |= (h1 ? 1u : 0u) << r; cm2 |= (h2 ? 1u : 0u) << r; cm3 |= (h3 ? 1u : 0u) << r;
                }
                const uint32_t in = pm & alive;   // rays that entered this node and are still undecided
                uint32_t nxt = 0;
                auto take = [&](uint32_t cm, int ref) {
                    cm &= in;
                    if (cm == 0u) return;
                    uint32_t e = pk_entry(cm, ref);
                    if (pk_is_leaf(e) && leafE == 0u) { leafE = e; return; }
                    if (nxt == 0u) { nxt = e; return; }
                    if (!pk_is_leaf(e) && pk_is_leaf(nxt)) { const uint32_t t = nxt; nxt = e; e = t; }   // go on with the inner node, defer the leaf
                    stk[sp * 64] = e;
                    sp++;
                };
                take(cm0, (int)f2u(q6.x)); take(cm1, (int)f2u(q6.y)); take(cm2, (int)f2u(q6.z)); take(cm3, (int)f2u(q6.w));
                if (nxt == 0u) pop_or_finish();
                else cur = nxt;
            }
        }
        // ---- leaves
        if (active && leafE != 0u) {
            const uint32_t v = leafE & 0x07ffffffu;
            const int first = (int)(v >> 3), count = (int)(v & 7u) + 1;
            uint32_t m = (leafE >> 28) & alive;
            auto test_tri = [&](V3 v0, V3 e1, V3 e2) {
                // tri_hit (rt_bvh.glsl:154-170) for every ray of the leaf's mask; tvec, qvec and dot(e2, qvec) do not depend on the direction
                const V3 tvec = ro - v0;
                const V3 qvec = cross(tvec, e1);
                const float te = dot(e2, qvec);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const V3 pvec = cross(rd[r], e2);
                    const float det = dot(e1, pvec);
                    const float invDet = 1.0f / det;
                    const float u = dot(tvec, pvec) * invDet;
                    const float vv = dot(rd[r], qvec) * invDet;
                    const float tt = te * invDet;
                    const bool hit = !(__builtin_fabsf(det) < 1e-8f) && !(u < 0.0f || u > 1.0f) && !(vv < 0.0f || u + vv > 1.0f) && !(tt < eps || tt > tMax[r]);
                    if (hit && ((m >> r) & 1u)) { occl |= 1u << r; alive &= ~(1u << r); m &= ~(1u << r); }
                }
            };
            for (int i = 0; i + 2 <= count && m != 0u; i += 2) {
                const v4f *tv = reinterpret_cast<const v4f *>(pairs + (size_t)(first + (i >> 1)) * 5);
                gathers += 5u;
                v4f r0 = tv[0], r1 = tv[1], r2 = tv[2], r3 = tv[3], r4 = tv[4];
                pin(r0); pin(r1); pin(r2); pin(r3); pin(r4);
                test_tri(mk3(r0.x, r0.y, r0.z), mk3(r0.w, r1.x, r1.y), mk3(r1.z, r1.w, r2.x));
                if (m != 0u) test_tri(mk3(r2.y, r2.z, r2.w), mk3(r3.x, r3.y, r3.z), mk3(r3.w, r4.x, r4.y));
            }
            if ((count & 1) && m != 0u) {
                const v4f *tv = reinterpret_cast<const v4f *>(pairs + (size_t)(first + (count >> 1)) * 5);
                gathers += 3u;
                v4f r0 = tv[0], r1 = tv[1], r2 = tv[2];
                pin(r0); pin(r1); pin(r2);
                test_tri(mk3(r0.x, r0.y, r0.z), mk3(r0.w, r1.x, r1.y), mk3(r1.z, r1.w, r2.x));
            }
            leafE = 0;
            if (alive == 0u) retire();
            else {
                if (cur != 0u && pk_is_leaf(cur)) { leafE = cur; cur = 0; }   // a second leaf was waiting in `cur`
                if (cur == 0u) pop_or_finish();
            }
        }
    }
    if (tally) {
        unsigned long long t = traced;
        for (int off = 32; off > 0; off >>= 1) t += __shfl_down(t, off, 64);
        if (lane == 0 && t) atomicAdd(tally, t);
    }
    if (gatherLoads) {
        unsigned long long g = gathers;
        for (int off = 32; off > 0; off >>= 1) g += __shfl_down(g, off, 64);
        if (lane == 0 && g) atomicAdd(gatherLoads, g);
    }
}

// ---- stage: post_primary -------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_post_primary(const DevFrame *__restrict__ fr, Targets tg, WaveBuf wb) {   // workgroup = kAppendBatch x 256 candidates
    const uint32_t n = wb.counts[0];
    RT_BATCH_APPEND(ap);
    for (int k = 0; k < kAppendBatch; ++k) {
        const uint32_t i = (blockIdx.x * kAppendBatch + k) * 256 + threadIdx.x;
        bool hit = false;
        if (i < n) {
            hit = wb.primTri[i] >= 0;
            if (!hit) {
                const uint32_t slot = wb.cand[i];
                int px, py;
                slot_to_pixel(fr->g, slot, px, py);
                finish_miss(fr, wb, (int)slot, px, py, primaryDirK(fr, sub_frame_of_slot(fr->g, slot), px, py));
            }
        }
        ap.note(k, hit);
    }
    ap.commit(&wb.counts[1]);
    for (int k = 0; k < kAppendBatch; ++k) {
        const uint32_t idx = ap.index(k);
        if (ap.mine(k)) {
            const uint32_t i = (blockIdx.x * kAppendBatch + k) * 256 + threadIdx.x;
            HitRec h;
            h.slot = wb.cand[i]; h.t = wb.primT[i]; h.tri = wb.primTri[i];
            wb.hits[idx] = h;
        }
    }
}

// ---- tracer policies ---------------------------------------------------------------------------
RT_DEV float4 mkf4(V3 v, float w) { return make_float4(v.x, v.y, v.z, w); }
RT_DEV float below(float r) { return (r > 0.0f) ? u2f(f2u(r) - 1u) : -1.0f; }   // largest float < r (r > 0), else "no ray"

struct GenDirectTracer {   // records first-generation rays of (hit j, sample s)
    WaveBuf wb;
    uint32_t j;
    int s;
    uint32_t shadowMask;
    bool giCast;
    RT_DEV bool shadow(int, int k, V3 ro, V3 rd, float tMax, bool matters) {
        shadowMask |= 1u << k;
        // sun (k = 4) and point-light (k = 5) rays start at hp + N*e / hp + L*e towards a fixed light: the same ray for every
        // sample of the pixel (rt_lighting.glsl:114-214 never look at the seed) -- sample 0 traces it, the others reuse its answer
        if (k >= 4 && s > 0) return false;
        uint32_t a = wb.sh1_slot(s, k) * wb.CH + j;
        if (!matters) { wb.shT[a] = -1.0f; return false; }   // dead ray: its answer is multiplied by zero
        wb.shT[a] = fmaxr(tMax, 0.0f);
        wb.shO[a] = mkf4(ro, 0.0f);
        wb.shD[a] = mkf4(rd, 0.0f);
        return false;
    }
    V3 giRo, giRd;   // RT_BIN_GI: the bounce ray is kept here and written by the workgroup's sort (k_gen_direct)
    RT_DEV int gi(V3 ro, V3 rd, V3 &, V3 &) {
        giCast = true;
        if (wb.giPerm) { giRo = ro; giRd = rd; return -1; }
        uint32_t a = (uint32_t)s * wb.CH + j;
        wb.giL[a] = 1.0f;
        wb.giO[a] = mkf4(ro, 0.0f);
        wb.giD[a] = mkf4(rd, 0.0f);
        return -1;
    }
    RT_DEV bool ao(int i, V3 org, V3 dir, float radius) {
        uint32_t a = (uint32_t)i * wb.CH + j;
        wb.shT[a] = below(radius);   // closest t < radius  <=>  any hit with t <= pred(radius)
        wb.shO[a] = mkf4(org, 0.0f);
        wb.shD[a] = mkf4(dir, 0.0f);
        return false;
    }
};
struct GenGiTracer {       // reads the bounce result, records the shadow rays at the bounce hit
    WaveBuf wb;
    const DevScene *sc;
    float inf;
    uint32_t j;
    int s;
    uint32_t pos;
    uint32_t shadowMask;
    RT_DEV bool shadow(int seg, int k, V3 ro, V3 rd, float tMax, bool matters) {
        if (seg != SEG_GI_DIRECT) return false;
        if (pos >= wb.q2Stride) return false;          // beyond the queue's capacity: k_gen_gi_overflow traces this pair's rays itself
        uint32_t a = (uint32_t)k * wb.q2Stride + pos;
        shadowMask |= 1u << k;
        if (!matters) { wb.sh2T[a] = -1.0f; return false; }
        wb.sh2T[a] = fmaxr(tMax, 0.0f);
        wb.sh2O[a] = mkf4(ro, 0.0f);
        wb.sh2D[a] = mkf4(rd, 0.0f);
        return false;
    }
    RT_DEV int gi(V3 ro, V3 rd, V3 &hp, V3 &hn) {
        uint32_t a = wb.gi_entry(s, j);
        int tri = wb.giTri[a];
        if (tri < 0) return 0;
        hp = ro + rd * wb.giT[a];
        hn = tri_normal(*sc, tri);
        return 1;
    }
    RT_DEV bool ao(int, V3, V3, float) { return false; }
};
struct CombineTracer {     // reads everything
    WaveBuf wb;
    const DevScene *sc;
    uint32_t j;
    int s;
    RT_DEV bool shadow(int seg, int k, V3, V3, float, bool matters) {
        if (!matters) return false;
        if (seg == SEG_DIRECT) return wb.occ1[wb.sh1_slot(k >= 4 ? 0 : s, k) * wb.CH + j] != 0;   // sun / point: sample 0's ray
        const uint32_t gp = (uint32_t)wb.giPos[(uint32_t)s * wb.CH + j];
        return (gp < wb.q2Stride ? wb.occ2[(uint32_t)k * wb.q2Stride + gp] : wb.occOvf[(uint32_t)k * (wb.CH * (uint32_t)wb.SPP) + gp]) != 0;
    }
    RT_DEV int gi(V3 ro, V3 rd, V3 &hp, V3 &hn) {
        uint32_t a = wb.gi_entry(s, j);
        int tri = wb.giTri[a];
        if (tri < 0) return 0;
        hp = ro + rd * wb.giT[a];
        hn = tri_normal(*sc, tri);
        return 1;
    }
    RT_DEV bool ao(int i, V3, V3, float radius) { return radius > 0.0f && wb.occ1[(uint32_t)i * wb.CH + j] != 0; }
};

struct HitCtx { Frag F; V3 dir, hp, hn; int px, py; uint32_t slot; };
RT_DEV HitCtx load_hit(const DevFrame *fr, const HitRec &h) {
    HitCtx c;
    c.slot = h.slot;
    slot_to_pixel(fr->g, h.slot, c.px, c.py);
    c.F.u = &fr->u; c.F.sc = &fr->sc; c.F.fcx = (float)c.px + 0.5f; c.F.fcy = (float)c.py + 0.5f;
    const int k = sub_frame_of_slot(fr->g, h.slot);
    c.F.frameIndex = fr->u.frameIndex + k;
    c.dir = primaryDirK(fr, k, c.px, c.py);
    c.hp = ld3(fr->u.camPos) + c.dir * h.t;
    c.hn = tri_normal(fr->sc, h.tri);
    return c;
}
// live hits of the chunk starting at c0: |[c0, c0+CH) ∩ [0, hits)|, written without a wrapping subtraction (hipcc -O3 was
// seen to drop the `h > c0 ? ... : 0` guard of the obvious form, turning empty chunks into full ones)
RT_DEV uint32_t chunk_live(const WaveBuf &wb, uint32_t c0) { uint32_t h = wb.counts[1]; return min(h, c0 + wb.CH) - min(h, c0); }

// ---- stage: gen_direct  (thread = (hit j, sample s), s-major so a wave shares s) -----------------
__global__ __launch_bounds__(256) void k_gen_direct(const DevFrame *__restrict__ fr, WaveBuf wb, uint32_t c0) {
    const RtUniforms &u = fr->u;
    const uint32_t live = chunk_live(wb, c0);
    const uint32_t tid = blockIdx.x * 256 + threadIdx.x;
    const bool mine = live != 0 && tid < live * (uint32_t)wb.SPP;
    if (!mine && (!wb.giPerm || live == 0 || blockIdx.x * 256u >= live * (uint32_t)wb.SPP)) return;   // (the sort below needs whole workgroups)
    const int s = mine ? (int)(tid / live) : 0;
    const uint32_t j = mine ? tid % live : 0;
    GenDirectTracer tr;
    tr.wb = wb; tr.j = j; tr.s = s; tr.shadowMask = 0; tr.giCast = false;
    tr.giRo = mk3(0.0f); tr.giRd = mk3(0.0f);
    if (mine) {
        HitCtx c = load_hit(fr, wb.hits[c0 + j]);
        const int SPP = max(u.spp, 1);
        const int seed = (int)((uint32_t)c.F.frameIndex * (uint32_t)SPP + (uint32_t)s);
        (void)directLightBVH(tr, c.F, SEG_DIRECT, c.hp, c.hn, seed, -c.dir);
        if (s == 0)
            for (int k = 4; k < 6; ++k)   // sun / point rays are conditional (rt_lighting.glsl:123,194)
                if (!(tr.shadowMask & (1u << k))) wb.shT[wb.sh1_slot(0, k) * wb.CH + j] = -1.0f;
        Work w;
        if (u.enableGI == 1) (void)oneBounceGIBVH<GenDirectTracer, false>(tr, c.F, c.hp, c.hn, c.F.frameIndex, seed, w);
        if (!tr.giCast && !wb.giPerm) wb.giL[(uint32_t)s * wb.CH + j] = -1.0f;
        if (s == 0 && wb.A > 0) (void)computeAO_BVH(tr, c.F, c.hp, c.hn, c.F.frameIndex);
    }
    if (wb.giPerm) {
        // EXPERIMENT (RT_BIN_GI=1): the workgroup's 256 bounce rays -- 256 consecutive hits of one sample, i.e. neighbouring
        // pixels -- are written sorted by direction (octant, then two bits of each |component|), so that quad-mates of the bounce launch start out
        // with like rays.  Results stay addressable through giPerm; nothing else changes, frames are bit-identical.
        __shared__ uint32_t sKey[256], sAddr[256];
        uint32_t key = 0xffffffffu;
        if (mine && tr.giCast) {
            const V3 d = tr.giRd;
            const uint32_t oct = (d.x < 0.0f ? 1u : 0u) | (d.y < 0.0f ? 2u : 0u) | (d.z < 0.0f ? 4u : 0u);
            auto q2 = [](float v) { return (uint32_t)min(3, (int)(__builtin_fabsf(v) * 4.0f)); };
            key = (oct << 6) | (q2(d.x) << 4) | (q2(d.y) << 2) | q2(d.z);
        }
        sKey[threadIdx.x] = key;
        sAddr[threadIdx.x] = mine ? (uint32_t)s * wb.CH + j : 0xffffffffu;
        __syncthreads();
        uint32_t rank = 0;
        for (uint32_t i = 0; i < 256u; ++i) { const uint32_t ki = sKey[i]; rank += (ki < key || (ki == key && i < threadIdx.x)) ? 1u : 0u; }
        // thread t's record goes to the address of the rank-th thread; threads outside the list (not mine) have the largest keys AND the largest
        // thread indices, so the first `mine` ranks map onto the `mine` addresses
        const uint32_t target = sAddr[rank];
        if (mine) {
            wb.giPerm[(uint32_t)s * wb.CH + j] = (int)target;
            if (tr.giCast) { wb.giL[target] = 1.0f; wb.giO[target] = mkf4(tr.giRo, 0.0f); wb.giD[target] = mkf4(tr.giRd, 0.0f); }
            else wb.giL[target] = -1.0f;
        }
    }
}

// ---- stage: gen_gi -------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_gen_gi(const DevFrame *__restrict__ fr, WaveBuf wb, uint32_t c0, uint32_t *giCount) {
    const RtUniforms &u = fr->u;
    const uint32_t live = chunk_live(wb, c0);
    const uint32_t tid = blockIdx.x * 256 + threadIdx.x;
    const bool mine = live != 0 && tid < live * (uint32_t)wb.SPP;
    const int s = mine ? (int)(tid / live) : 0;
    const uint32_t j = mine ? tid % live : 0;
    const uint32_t a = (uint32_t)s * wb.CH + j;
    const uint32_t ae = mine ? wb.gi_entry(s, j) : 0u;   // where this (hit, sample)'s bounce ray and its answer are
    const bool bounced = mine && wb.giL[ae] >= 0.0f && wb.giTri[ae] >= 0;
    const uint32_t pos = block_append(bounced, giCount);   // compact the (hit, sample) pairs that need second-generation rays
    if (!mine) return;
    wb.giPos[a] = bounced ? (int)pos : -1;
    if (!bounced) return;
    GenGiTracer tr;
    tr.wb = wb; tr.sc = &fr->sc; tr.inf = u.inf; tr.j = j; tr.s = s; tr.pos = pos; tr.shadowMask = 0;
    HitCtx c = load_hit(fr, wb.hits[c0 + j]);
    const int SPP = max(u.spp, 1);
    const int seed = (int)((uint32_t)c.F.frameIndex * (uint32_t)SPP + (uint32_t)s);
    Work w;
    (void)oneBounceGIBVH<GenGiTracer, false>(tr, c.F, c.hp, c.hn, c.F.frameIndex, seed, w);
    if (pos < wb.q2Stride)
        for (int k = 0; k < 6; ++k)
            if (!(tr.shadowMask & (1u << k))) wb.sh2T[(uint32_t)k * wb.q2Stride + pos] = -1.0f;
}

// Shadow queue 2 of a large launch set holds a PREDICTED number of bounce hits (rt_wave_render).  The (hit, sample) pairs beyond it -- none, unless the view changed so
// that more than twice as many bounce rays hit as in any batch before -- get their six shadow rays traced right here, one thread per pair, with the megakernel's any-hit
// walk (bvh_anyhit: the same answers as the any-hit launch, tests/test_gpu_parity.py), into occOvf.  Launched behind every k_gen_gi of such a set; returns at once when
// nothing overflowed.
struct GenGiOverflowTracer {
    WaveBuf wb;
    const DevScene *sc;
    float eps;
    StackEntry *stk;
    uint32_t j;
    int s;
    uint32_t pos;
    RT_DEV bool shadow(int seg, int k, V3 ro, V3 rd, float tMax, bool matters) {
        if (seg != SEG_GI_DIRECT) return false;
        bool occ = false;
        if (matters) { Work w; occ = bvh_anyhit<false>(*sc, ro, rd, eps, fmaxr(tMax, 0.0f), stk, w); }
        wb.occOvf[(uint32_t)k * (wb.CH * (uint32_t)wb.SPP) + pos] = occ ? 1 : 0;
        return false;
    }
    RT_DEV int gi(V3 ro, V3 rd, V3 &hp, V3 &hn) {
        uint32_t a = wb.gi_entry(s, j);
        int tri = wb.giTri[a];
        if (tri < 0) return 0;
        hp = ro + rd * wb.giT[a];
        hn = tri_normal(*sc, tri);
        return 1;
    }
    RT_DEV bool ao(int, V3, V3, float) { return false; }
};
__global__ __launch_bounds__(256) void k_gen_gi_overflow(const DevFrame *__restrict__ fr, WaveBuf wb, uint32_t c0, const uint32_t *giCount, int stackEntries) {
    if (*giCount <= wb.q2Stride) return;      // the normal case: a small fixed grid that leaves at once (a grid of one thread per pair -- 29 000 workgroups for a batch of eight
                                              // 1080p frames -- cost 2 % of a frame just to be dispatched and return)
    const uint32_t live = chunk_live(wb, c0);
    if (live == 0) return;
    const uint32_t n = live * (uint32_t)wb.SPP;
    for (uint32_t tid = blockIdx.x * 256 + threadIdx.x; tid < n; tid += gridDim.x * 256) {
        const int s = (int)(tid / live);
        const uint32_t j = tid % live;
        const int gp = wb.giPos[(uint32_t)s * wb.CH + j];
        if (gp < 0 || (uint32_t)gp < wb.q2Stride) continue;
        GenGiOverflowTracer tr;
        tr.wb = wb; tr.sc = &fr->sc; tr.eps = fr->u.eps; tr.j = j; tr.s = s; tr.pos = (uint32_t)gp;
        tr.stk = reinterpret_cast<StackEntry *>(rt_dyn_lds) + (threadIdx.x >> 6) * stackEntries * 64 + (threadIdx.x & 63);
        HitCtx c = load_hit(fr, wb.hits[c0 + j]);
        const int SPP = max(fr->u.spp, 1);
        const int seed = (int)((uint32_t)c.F.frameIndex * (uint32_t)SPP + (uint32_t)s);
        Work w;
        (void)oneBounceGIBVH<GenGiOverflowTracer, false>(tr, c.F, c.hp, c.hn, c.F.frameIndex, seed, w);
    }
}

// ---- stage: combine (thread = hit) ---------------------------------------------------------------
__global__ __launch_bounds__(256) void k_combine(const DevFrame *__restrict__ fr, Targets tg, WaveBuf wb, uint32_t c0) {
    const RtUniforms &u = fr->u;
    const uint32_t live = chunk_live(wb, c0);
    const uint32_t j = blockIdx.x * 256 + threadIdx.x;
    if (j >= live) return;
    HitCtx c = load_hit(fr, wb.hits[c0 + j]);
    const int SPP = max(u.spp, 1);
    CombineTracer tr;
    tr.wb = wb; tr.sc = &fr->sc; tr.j = j; tr.s = 0;
    Work w;
    V2 prevNDC = ndcFromWorld(c.hp, u.prevViewProj), currNDC = ndcFromWorld(c.hp, u.currViewProj);
    V2 motionOut = mk2(currNDC.x - prevNDC.x, currNDC.y - prevNDC.y);
    V3 nn = normalize(c.hn);
    float ao = 1.0f;
    if (u.enableAO == 1) ao = computeAO_BVH(tr, c.F, c.hp, c.hn, c.F.frameIndex);
    V3 frameSum = mk3(0.0f);
    for (int s = 0; s < SPP; ++s) {
        tr.s = s;
        int seed = (int)((uint32_t)c.F.frameIndex * (uint32_t)SPP + (uint32_t)s);
        frameSum = frameSum + shadeSampleBVH<CombineTracer, false>(tr, c.F, c.hp, c.hn, -c.dir, seed, ao, w);
    }
    finish_pixel(fr, wb, (int)c.slot, frameSum, motionOut, mk4(c.hp.x, c.hp.y, c.hp.z, 1.0f), mk4(nn.x, nn.y, nn.z, 0.0f));
}

__global__ void k_accum_tally(const uint32_t *counts, unsigned long long *acc, int frames) {
    // acc: [0] candidates [1] hits [2] primary rays traced [3] shadow [4] bounce [5] bounce-shadow (the traversal kernels add to
    // [2..5] themselves) [6] frames
    int i = threadIdx.x;
    if (i < 2) acc[i] += counts[i];
    if (i == 6) acc[6] += (unsigned long long)frames;
}

template <class Src, bool ANY>
void launch_trace(hipStream_t st, int cus, int gridPct, int depth, const DevFrame *fr, const DevScene &hs, Src src, uint32_t *head, unsigned long long *tally,
                  unsigned long long *gatherLoads, TraceTune tune, unsigned long long *stats = nullptr) {
    // Stack entries: closest-hit defers one sibling per binary level (8 B each); any-hit walks 4-wide nodes and can
    // defer three per two levels (4 B each).  Resident 256-thread workgroups per CU follow from the LDS footprint and the kernel's
    // registers: asked from the runtime per (kernel, stack size), the persistent grid is exactly what fits.
    const int stack = std::max(4, ANY ? (hs.anyStack > 0 ? hs.anyStack : 3 * ((depth + 1) / 2)) : depth);
    const size_t ldsBytes = (size_t)256 * stack * (ANY ? 4 : 8);
    const bool qn = ANY && tune.qnodes != 0 && hs.q4 != nullptr && !stats && !tune.nearFirst;   // (the diagnostic and near-first builds walk the exact nodes)   // -1: whenever rt_upload_bvh built the quantised nodes (trees beyond the L2)
    const bool fuse = !ANY && tune.fused != 0 && hs.wF != nullptr && !tune.coop;   // closest-hit launches: the fused records when rt_upload_bvh built them
    const bool impl = tune.impl != 0 && (!stats || tune.timing) && (ANY ? (hs.iN4 != nullptr && (!qn || hs.iQ4 != nullptr) && !tune.nearFirst && tune.leafb < 4 && !(qn && tune.qnodes == 1) && hs.anyStack > 0)
                                                       : (!fuse && hs.iN2 != nullptr && !tune.coop));   // ... the implicit records when every leaf sits at one depth
    const float4 *nodes = ANY ? (impl ? (qn ? hs.iQ4 : hs.iN4) : (qn ? hs.q4 : hs.w4)) : (fuse ? hs.wF : (impl ? hs.iN2 : hs.wnodesW));
    const float4 *pairRecords = impl ? hs.iPairs : hs.pairs;
    const float4 *leafBoxes = (ANY && impl) ? hs.iLeafBox : hs.leafBox;
    auto go = [&](auto kernel) {
        // the runtime's answer per (device, kernel, LDS bytes): a process may hold contexts on devices of different shapes
        thread_local std::map<std::tuple<int, const void *, size_t>, int> occ;
        int dev = 0;
        (void)hipGetDevice(&dev);
        int &perCU = occ[std::make_tuple(dev, (const void *)kernel, ldsBytes)];
        if (perCU == 0) {
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCU, kernel, 256, ldsBytes) != hipSuccess || perCU < 1) perCU = 1;
            perCU = std::min(perCU, 8);
        }
        const unsigned blocks = (unsigned)std::max(8, cus * perCU * gridPct / 100);
        hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), ldsBytes, st, fr, nodes, pairRecords, src, head, tally, gatherLoads, tune, stack, stats, leafBoxes);
    };
    const int leafb = ANY ? tune.leafb : tune.leafbClosest;
    if (stats && tune.timing) {   // RT_TRACE_TIMING=1: the production kernel of this launch with scalar time stamps (exact / implicit records, default leaf groups only)
        if (impl) go(k_trace<Src, ANY, 2, false, false, false, 0, false, true, true>); else go(k_trace<Src, ANY, 2, false, false, false, 0, false, false, true>);
        return;
    }
    if (impl && ANY) { if (qn) go(k_trace<Src, ANY, 2, false, false, false, ANY ? 2 : 0, false, true>); else go(k_trace<Src, ANY, 2, false, false, false, 0, false, true>); }
    else if (impl) go(k_trace<Src, ANY, 2, false, false, false, 0, false, true>);
    else if (stats && fuse) go(k_trace<Src, ANY, 2, true, false, false, 0, !ANY>);
    else if (stats) { if (leafb >= 4) go(k_trace<Src, ANY, 4, true>); else go(k_trace<Src, ANY, 2, true>); }
    else if (fuse) go(k_trace<Src, ANY, 2, false, false, false, 0, !ANY>);
    else if (!ANY && tune.coop) go(k_trace<Src, ANY, 2, false, !ANY>);
    else if (ANY && tune.nearFirst) go(k_trace<Src, ANY, 2, false, false, ANY>);
    else if (qn && tune.qnodes == 1) go(k_trace<Src, ANY, 2, false, false, false, ANY ? 1 : 0>);   // seven waves per SIMD, 44 B of scratch: slower (measured)
    else if (qn) go(k_trace<Src, ANY, 2, false, false, false, ANY ? 2 : 0>);
    else       { if (leafb >= 4) go(k_trace<Src, ANY, 4, false>); else go(k_trace<Src, ANY, 2, false>); }
}

void launch_packets(hipStream_t st, int cus, int gridPct, int depth, const DevFrame *fr, const DevScene &hs, PacketSrc src, uint32_t *head, unsigned long long *tally,
                    unsigned long long *gatherLoads, TraceTune tune) {
    const int stack = std::max(4, hs.anyStack > 0 ? hs.anyStack : 3 * ((depth + 1) / 2));
    const size_t ldsBytes = (size_t)256 * stack * 4;
    thread_local std::map<std::pair<int, size_t>, int> occ;
    int dev = 0;
    (void)hipGetDevice(&dev);
    int &perCU = occ[{dev, ldsBytes}];
    if (perCU == 0) {
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCU, k_trace_packets, 256, ldsBytes) != hipSuccess || perCU < 1) perCU = 1;
        perCU = std::min(perCU, 8);
    }
    const unsigned blocks = (unsigned)std::max(8, cus * perCU * gridPct / 100);
    hipLaunchKernelGGL(k_trace_packets, dim3(blocks), dim3(256), ldsBytes, st, fr, hs.w4, hs.pairs, src, head, tally, gatherLoads, tune, stack);
}

}  // namespace

// -------------------------------------------------------------------------------------------------
constexpr int kMinLaunches = 256;   // trace launches per frame (1 + 3 per chunk) the cursor table starts with; grown on demand

#ifndef RT_MAX_LANES
#define RT_MAX_LANES 8
#endif
struct RtArenaPool {
    int n = 0;
    void *mem[RT_MAX_LANES] = {};
    size_t bytes[RT_MAX_LANES] = {};
    double q2Frac = 0.0;                       // largest share of (hit, sample) pairs whose bounce ray hit, over the launch sets of all lanes so far (read back with the NEXT
                                               // launch set's hit count: no extra synchronisation); 0: nothing known yet
    size_t maxCh[RT_MAX_LANES] = {};           // most hits an arena has been sized for
    hipEvent_t freeEv[RT_MAX_LANES] = {};      // recorded behind the last kernel of the batch that used the arena last
    hipStream_t lastUser[RT_MAX_LANES] = {};   // (a later batch on the same stream is ordered behind it anyway)
};
RtArenaPool *rt_arena_pool_create(int arenas) {
    RtArenaPool *p = new RtArenaPool();
    p->n = std::max(1, std::min(arenas, (int)RT_MAX_LANES));
    for (int i = 0; i < RT_MAX_LANES; ++i) (void)hipEventCreateWithFlags(&p->freeEv[i], hipEventDisableTiming);
    return p;
}
void rt_arena_pool_destroy(RtArenaPool *p) {
    if (!p) return;
    for (int i = 0; i < RT_MAX_LANES; ++i) { if (p->mem[i]) (void)hipFree(p->mem[i]); if (p->freeEv[i]) (void)hipEventDestroy(p->freeEv[i]); }
    delete p;
}
int rt_arena_pool_count(const RtArenaPool *p) { int c = 0; for (int i = 0; p && i < RT_MAX_LANES; ++i) c += p->mem[i] ? 1 : 0; return c; }
size_t rt_arena_pool_bytes(const RtArenaPool *p) { size_t b = 0; for (int i = 0; p && i < RT_MAX_LANES; ++i) b += p->bytes[i]; return b; }

struct RtWave {
    std::string err;
    RtArenaPool *pool = nullptr;   // null: this lane owns its arena (chunkArena below)
    int arena = 0;                 // index into pool: lane % n for a launch set of one chunk, the lane itself for a set of several (see rt_wave_render)
    int lane = 0;
    int cus = 256;
    // ray-queue budget per frame lane; 288 GB of HBM make this cheap.  16 GB hold the queues of a whole batch of eight 1080p / 4 spp frames (7.4 M hits x
    // 2.1 KB) in ONE chunk: no hit-count read-back, half the launches (1.76-1.80 -> 1.68-1.72 ms per frame against 8 GB; profiles/r03_experiments.txt)
    size_t budgetBytes = (size_t)16 << 30;
    TraceTune tune = default_tune();   // chunk 0 = run length chosen in the kernel from the queue size
    // allocations
    size_t slotsCap = 0;      // per-frame arrays sized for this many pixel slots
    size_t chunkBytes = 0;    // bytes of the per-chunk arena
    int prevChunks = 0, prevSPP = 1;   // chunks / spp of this lane's previous launch set: its hit count and bounce-hit counts are copied to hostHits[1], [2..] before the counters are cleared
    double q2FracOwn = 0.0;   // (no pool)
    void *frameArena = nullptr, *chunkArena = nullptr, *resultArena = nullptr;
    size_t resultBytes = 0;   // per-lane traversal results (what k_combine reads)
    uint32_t *counts = nullptr, *heads = nullptr;
    int launchCap = 0, chunkCap = 0;     // trace launches `heads` holds cursors for / chunks `counts` holds bounce counters for
    unsigned long long *acc = nullptr;   // traced-ray tallies accumulated over frames
    unsigned long long *stats = nullptr; // RT_TRACE_STATS=1: 4 stages x 8 diagnostic sums
    uint32_t *hostHits = nullptr;        // pinned: the hit count of a batch that needs more than one chunk (read back once per batch)
    bool packetAO = false;               // RT_PACKET_AO=1 (measured option, round 4): the AO rays of a hit traced as one packet (k_trace_packets) instead of one by one in the any-hit launch
    hipStream_t shadeStream = nullptr;   // RT_CU_SPLIT: the shading kernels' stream (a CU-masked one), else null
    hipEvent_t hopEv = nullptr;
    bool binGi = false;                  // RT_BIN_GI=1: bounce rays sorted by direction inside each workgroup of k_gen_direct (experiment)
    bool chunksFromSlots = false;        // RT_CHUNKS_FROM_SLOTS=1 (tests): launch the chunk loop for every pixel slot, as rounds 1-2 did
};

RtWave *rt_wave_create(int cus, RtArenaPool *pool, int lane) {
    RtWave *w = new RtWave();
    w->cus = cus > 0 ? cus : 256;
    w->pool = pool;
    w->lane = lane;
    w->arena = pool ? lane % pool->n : 0;
    if (const char *e = getenv("RT_QUEUE_BUDGET_MB")) w->budgetBytes = (size_t)atoll(e) << 20;
    if (const char *e = getenv("RT_REFILL_MIN")) w->tune.refillMin = std::max(1, std::min(64, atoi(e)));
    if (const char *e = getenv("RT_CHUNK")) { int v = atoi(e); w->tune.chunk = v <= 0 ? 0 : std::max(8, std::min(1 << 20, v)); }   // 0 = from the queue size
    if (const char *e = getenv("RT_DEBUG_SKIP_TRAVERSAL")) w->tune.skipTraversal = atoi(e);
    if (const char *e = getenv("RT_LEAFB")) w->tune.leafb = w->tune.leafbClosest = atoi(e);
    if (const char *e = getenv("RT_LEAFB_CLOSEST")) w->tune.leafbClosest = atoi(e);
    if (const char *e = getenv("RT_MIN_SEARCH")) w->tune.minSearch = std::max(0, std::min(64, atoi(e)));
    if (const char *e = getenv("RT_CHUNKS_FROM_SLOTS")) w->chunksFromSlots = atoi(e) != 0;
    if (const char *e = getenv("RT_QUAD_REFILL")) w->tune.quadRefill = atoi(e) != 0;
    if (const char *e = getenv("RT_COOP")) w->tune.coop = atoi(e);
    if (const char *e = getenv("RT_BIN_GI")) w->binGi = atoi(e) != 0;
    if (const char *e = getenv("RT_NEAR_FIRST")) w->tune.nearFirst = atoi(e);
    if (const char *e = getenv("RT_REVERSE")) w->tune.reverse = atoi(e);
    if (const char *e = getenv("RT_GUIDED")) w->tune.guided = atoi(e);
    if (const char *e = getenv("RT_DENSE_TAKE")) w->tune.denseTake = atoi(e);   // 0: every refill probes liveness first, as in rounds 2-3     // 0: runs of one length, as in rounds 1-3   // 0: any-hit queues dealt from their beginning, as in rounds 1-3
    if (const char *e = getenv("RT_PACKET_AO")) w->packetAO = atoi(e) != 0;
    if (const char *e = getenv("RT_QNODES")) w->tune.qnodes = atoi(e);        // 0: never; 1 / 2: always (seven / six waves per SIMD); default: when rt_upload_bvh built them
    (void)hipEventCreateWithFlags(&w->hopEv, hipEventDisableTiming);
    if (const char *e = getenv("RT_CU_SPLIT")) {
        const int k = std::max(1, std::min(7, atoi(e)));
        uint32_t mask[8];
        for (int i = 0; i < 8; ++i) { uint32_t m = 0; for (int b = 0; b < 32; ++b) if (((i * 32 + b) & 7) < k) m |= 1u << b; mask[i] = m; }
        if (hipExtStreamCreateWithCUMask(&w->shadeStream, 8, mask) != hipSuccess) w->shadeStream = nullptr;
    }   // quad-cooperative node fetch of the closest-hit launches (measured option)
    // EXPERIMENT RT_SHADE_PRIORITY=p: the shading kernels on a second stream of queue priority p (-1 high, 1 low; no CU mask), the traversal launches on the lane's own
    else if (const char *e = getenv("RT_SHADE_PRIORITY")) {
        int least = 0, greatest = 0;
        (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
        const int p = std::max(greatest, std::min(least, atoi(e)));
        if (hipStreamCreateWithPriority(&w->shadeStream, hipStreamNonBlocking, p) != hipSuccess) w->shadeStream = nullptr;
    }
    return w;
}
void rt_wave_destroy(RtWave *w) {
    if (!w) return;
    if (w->shadeStream) (void)hipStreamDestroy(w->shadeStream);
    if (w->hopEv) (void)hipEventDestroy(w->hopEv);
    if (w->frameArena) (void)hipFree(w->frameArena);
    if (w->chunkArena && !w->pool) (void)hipFree(w->chunkArena);
    if (w->resultArena) (void)hipFree(w->resultArena);
    if (w->counts) (void)hipFree(w->counts);
    if (w->heads) (void)hipFree(w->heads);
    if (w->acc) (void)hipFree(w->acc);
    if (w->stats) (void)hipFree(w->stats);
    if (w->hostHits) (void)hipHostFree(w->hostHits);
    delete w;
}
const char *rt_wave_error(const RtWave *w) { return w->err.c_str(); }
size_t rt_wave_frame_bytes(const RtWave *w) { return w ? w->slotsCap * (4 + 4 + 4 + sizeof(HitRec) + 16 + 4 + 8 + 8) + (w->pool ? 0 : w->chunkBytes) + w->resultBytes : 0; }

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

#define W_TRY(expr)                                                                   \
    do {                                                                              \
        hipError_t e_ = (expr);                                                       \
        if (e_ != hipSuccess) { w->err = std::string(#expr) + ": " + hipGetErrorString(e_); return RT_ERR_HIP; } \
    } while (0)

int rt_wave_render(RtWave *w, RtContext *ctx, hipStream_t st, const DevFrame *dFrame, const DevFrame &host, Targets tg,
                   unsigned long long *, bool count, int treeDepth, hipEvent_t evPrevDone, bool cacheResident) {
    if (count) { w->err = "work counters are produced by the megakernel pipeline (RT_PIPELINE_MEGAKERNEL)"; return RT_ERR_UNSUPPORTED; }
    const RtUniforms &u = host.u;
    const int batch = std::max(host.g.batch, 1);
    const size_t nSlots = (size_t)std::max(host.g.nLocalTiles, 1) * 256 * (size_t)batch;   // pixel slots of all frames of the batch
    const int SPP = std::max(u.spp, 1);
    const int A = (u.enableAO == 1) ? std::max(u.aoSamples, 0) : 0;
    const int S1 = A + 4 * SPP + 2, S2 = 6 * SPP;   // WaveBuf::sh1_slot

    if (!w->acc) { W_TRY(hipMalloc(&w->acc, 16 * sizeof(unsigned long long))); W_TRY(hipMemsetAsync(w->acc, 0, 16 * sizeof(unsigned long long), st)); }
    // per-frame arena: cand, primT, primTri, hits
    if (w->slotsCap < nSlots) {
        if (w->frameArena) (void)hipFree(w->frameArena);
        w->frameArena = nullptr;
        W_TRY(hipMalloc(&w->frameArena, nSlots * (4 + 4 + 4 + sizeof(HitRec) + 16 + 4 + 8 + 8)));
        w->slotsCap = nSlots;
    }
    // chunk capacity from the budget
    const size_t perHit = (size_t)(S1 + SPP + S2) * 36 + (size_t)S1 + (size_t)SPP * 12 + (size_t)S2;
    const size_t CHbudget = align_up(std::min(nSlots, std::max<size_t>(w->budgetBytes / perHit, 4096)), 256);
    // ray records + liveness words (read by the traversal launches only: a SHARED arena can be handed on as soon as the last of them is done) ...
    // (shadow queue 2 -- six slots per bounce HIT -- is an allocation of its own since round 5: q2_bytes(entries per slot))
    auto rays_bytes = [&](size_t ch) {
        return align_up(ch * (size_t)S1 * 32, 256) + align_up(ch * (size_t)SPP * 32, 256) +
               align_up(ch * (size_t)S1 * 4, 256) + align_up(ch * (size_t)SPP * 4, 256) + 4096;
    };
    auto q2_bytes = [&](size_t n) { return align_up(n * 6 * 16, 256) * 2 + align_up(n * 6 * 4, 256) + 4096; };
    // ... and the results k_combine reads (1 byte per any-hit ray, 8 per bounce ray, 4 per (hit, sample)): the lane's own
    auto result_bytes = [&](size_t ch) {
        return align_up(ch * (size_t)S1, 256) + align_up(ch * (size_t)SPP * 8, 256) + align_up(ch * (size_t)S2, 256) * 2 + align_up(ch * (size_t)SPP * 4, 256) * 2 + 4096;
    };
    // arrays that hold `ch` hits (grown, never shrunk; `room`: allocate for that many when growing, so that a batch with a few more hits fits too)
    // Shadow queue 2 (round 5): six ray records per bounce HIT, behind queue 1 in the same allocation.  Sized for the worst case -- every bounce ray
    // hits -- it was 48 % of the ray arenas; on the bench view 1 % of that is used.  A LARGE launch set (sized from its hit count anyway) sizes it from what earlier
    // launch sets needed: twice the largest share of (hit, sample) pairs that bounced so far + slack, the worst case while nothing is known; pairs beyond the capacity -- a
    // view whose bounce hits more than doubled -- are traced in place by k_gen_gi_overflow instead of queued (an order of magnitude slower per ray: 1 M-triangle scene, 20 ms
    // per frame with half the pairs overflowing -- hence never a guess below the worst case).  Small launch sets take the worst case, no question asked of the device.
    const double &q2Share = w->pool ? w->pool->q2Frac : w->q2FracOwn;   // (live: updated behind this set's hit-count read-back, before a large set sizes its arena)
    bool predictQ2 = false;          // set for deferred sets below
    const bool predictOn = !(getenv("RT_Q2_PREDICT") && atoi(getenv("RT_Q2_PREDICT")) == 0);   // RT_Q2_PREDICT=0: always the worst case, as in rounds 1-4 (A/B)
    auto n2_of = [&](size_t ch) -> size_t {
        const size_t worst = ch * (size_t)SPP;
        size_t n2 = (predictOn && predictQ2 && q2Share > 0.0) ? std::max<size_t>((size_t)(2.0 * q2Share * (double)worst) + 65536, worst / 32) : worst;
        if (const char *e = getenv("RT_Q2_CAP")) n2 = std::max<size_t>((size_t)atoll(e), 64);   // tests: force the overflow path
        return align_up(std::min(n2, worst), 64);
    };
    auto arena_bytes = [&](size_t ch) { return rays_bytes(ch) + q2_bytes(n2_of(ch)); };
    auto ensure = [&](size_t ch, size_t room) -> int {
        room = std::max(room, ch);
        if (w->resultBytes < result_bytes(ch)) {
            if (w->resultArena) { W_TRY(hipStreamSynchronize(st)); (void)hipFree(w->resultArena); }
            w->resultArena = nullptr;
            w->resultBytes = 0;
            W_TRY(hipMalloc(&w->resultArena, result_bytes(room)));
            w->resultBytes = result_bytes(room);
        }
        if (w->pool) {
            // a shared arena: grown once its current user is done with it
            RtArenaPool &P = *w->pool;
            const int a = w->arena;
            // (grown when too small; given back ONCE it is more than 1.6 x what a launch set asks for now that the share of bounce hits is known: the worst-case
            // allocation of the first launch sets.  Either way behind the arena's current user.)
            // The new size is for the LARGEST launch set the arena has held, not for this one: bench.py's five-frame warm-up batch once shrank an arena that the
            // seven-frame batches of the timed region then had to grow again (a device-wide synchronisation and a multi-gigabyte hipMalloc inside a 34 ms region:
            // 1.65 -> 1.81 ms per step).
            room = std::max(room, P.maxCh[a]);
            if (P.bytes[a] < arena_bytes(ch) || (predictOn && predictQ2 && q2Share > 0.0 && P.bytes[a] > arena_bytes(room) + arena_bytes(room) / 2 + arena_bytes(room) / 10)) {
                if (P.lastUser[a]) W_TRY(hipEventSynchronize(P.freeEv[a]));
                if (P.mem[a]) (void)hipFree(P.mem[a]);
                P.mem[a] = nullptr; P.bytes[a] = 0;
                W_TRY(hipMalloc(&P.mem[a], arena_bytes(room)));
                P.bytes[a] = arena_bytes(room);
            }
            P.maxCh[a] = room;
            w->chunkArena = P.mem[a];
        } else if (w->chunkBytes < arena_bytes(ch)) {
            if (w->chunkArena) { W_TRY(hipStreamSynchronize(st)); (void)hipFree(w->chunkArena); }
            w->chunkArena = nullptr;
            w->chunkBytes = 0;
            W_TRY(hipMalloc(&w->chunkArena, arena_bytes(room)));
            w->chunkBytes = arena_bytes(room);
        }
        return RT_OK;
    };
    // Round 4: a launch set whose queues would be large if every pixel slot were a hit (a batch of eight 1080p frames: 30 GB) is sized from its HIT count --
    // read back once, behind k_post_primary, on this lane's stream only (the other lanes keep the GPU busy: 1.726 vs 1.719 ms per frame with and without
    // that read-back, profiles/r04_experiments.txt) -- instead of from the budget.  Small launch sets (and RT_CHUNKS_FROM_SLOTS) are sized from their pixel
    // slots as before and never wait for the host.
    constexpr size_t kComfortBytes = (size_t)4 << 30;
    const bool deferred = !w->chunksFromSlots && rays_bytes(CHbudget) + q2_bytes(CHbudget * (size_t)SPP) > kComfortBytes;
    size_t CH = CHbudget;
    WaveBuf wb;
    auto carve = [&](size_t ch) {
        char *q = (char *)w->chunkArena;
        auto take = [&](size_t bytes) { char *r = q; q += align_up(bytes, 256); return r; };
        wb.shO = (float4 *)take(ch * (size_t)S1 * 16); wb.shD = (float4 *)take(ch * (size_t)S1 * 16);
        wb.giO = (float4 *)take(ch * (size_t)SPP * 16); wb.giD = (float4 *)take(ch * (size_t)SPP * 16);
        wb.shT = (float *)take(ch * (size_t)S1 * 4); wb.giL = (float *)take(ch * (size_t)SPP * 4);
        const size_t n2 = n2_of(ch);
        wb.sh2O = (float4 *)take(n2 * 6 * 16); wb.sh2D = (float4 *)take(n2 * 6 * 16); wb.sh2T = (float *)take(n2 * 6 * 4);
        wb.q2Stride = (uint32_t)n2;
        q = (char *)w->resultArena;
        wb.occ1 = (uint8_t *)take(ch * (size_t)S1);
        wb.giT = (float *)take(ch * (size_t)SPP * 4); wb.giTri = (int *)take(ch * (size_t)SPP * 4);
        wb.occ2 = (uint8_t *)take(ch * (size_t)S2);
        wb.occOvf = (uint8_t *)take(ch * (size_t)S2);
        wb.giPos = (int *)take(ch * (size_t)SPP * 4);
        wb.giPerm = w->binGi ? (int *)take(ch * (size_t)SPP * 4) : nullptr;
        wb.CH = (uint32_t)ch;
    };
    if (CHbudget * (size_t)std::max(S1, S2) >= ((size_t)1 << 31)) { w->err = "ray queue chunk exceeds 2^31 entries; lower RT_QUEUE_BUDGET_MB"; return RT_ERR_UNSUPPORTED; }
    {
        char *p = (char *)w->frameArena;
        wb.cand = (uint32_t *)p; p += nSlots * 4;
        wb.primT = (float *)p; p += nSlots * 4;
        wb.primTri = (int *)p; p += nSlots * 4;
        wb.hits = (HitRec *)p; p += nSlots * sizeof(HitRec);
        wb.pendC = (float4 *)p; p += nSlots * 16;     // nSlots is a multiple of 256: every sub-array stays 16-byte aligned
        wb.pendPos = (uint2 *)p; p += nSlots * 8;
        wb.pendNrm = (uint2 *)p; p += nSlots * 8;
        wb.pendMy = (float *)p;
    }
    wb.shO = wb.shD = wb.giO = wb.giD = wb.sh2O = wb.sh2D = nullptr; wb.shT = wb.giL = wb.sh2T = nullptr;   // (deferred: carved behind k_post_primary)
    wb.occ1 = wb.occ2 = wb.occOvf = nullptr; wb.giT = nullptr; wb.giTri = wb.giPos = wb.giPerm = nullptr;
    wb.q2Stride = 0;
    if (!deferred) {
        if (w->pool) w->arena = (nSlots + CH - 1) / CH > 1 ? w->lane : w->lane % w->pool->n;
        int rc = ensure(CH, CH);
        if (rc != RT_OK) return rc;
        carve(CH);
    }
    wb.CH = (uint32_t)CH; wb.A = A; wb.SPP = SPP;
    int nChunks = (int)((nSlots + CH - 1) / CH);   // upper bound (every pixel slot a hit); cut down to the hit count below
    // cursor table (one set of sharded cursors per trace launch) and per-chunk bounce counters: grown when a small queue budget
    // cuts the frame into more chunks than seen so far (4K / 16 spp at RT_QUEUE_BUDGET_MB=128 is 491 chunks).  Both arrays are
    // only touched by this lane's stream, which is drained first.
    if (1 + nChunks * 4 > w->launchCap || nChunks > w->chunkCap) {
        W_TRY(hipStreamSynchronize(st));
        if (w->counts) (void)hipFree(w->counts);
        if (w->heads) (void)hipFree(w->heads);
        w->counts = w->heads = nullptr;
        w->launchCap = w->chunkCap = 0;
        w->prevChunks = 0;
        const int lc = std::max(kMinLaunches, 1 + nChunks * 4), cc = std::max(4096, nChunks);
        W_TRY(hipMalloc(&w->counts, (size_t)(64 + cc) * sizeof(uint32_t)));
        W_TRY(hipMalloc(&w->heads, (size_t)lc * kHeadWords * sizeof(uint32_t)));
        w->launchCap = lc; w->chunkCap = cc;
    }
    wb.counts = w->counts; wb.heads = w->heads;

    // the bounce-hit counts of this lane's PREVIOUS launch set, before they are cleared: read on the host behind this set's own hit-count read-back (no extra
    // synchronisation) -- what the capacity of shadow queue 2 is predicted from
    if (!w->hostHits) W_TRY(hipHostMalloc((void **)&w->hostHits, 64 * sizeof(uint32_t)));
    const int giCopied = std::min(w->prevChunks, 60);
    if (giCopied > 0) {
        W_TRY(hipMemcpyAsync(w->hostHits + 1, w->counts + 1, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        W_TRY(hipMemcpyAsync(w->hostHits + 2, w->counts + 64, (size_t)giCopied * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    }
    W_TRY(hipMemsetAsync(w->counts, 0, (size_t)(64 + nChunks) * sizeof(uint32_t), st));
    W_TRY(hipMemsetAsync(w->heads, 0, (size_t)(1 + nChunks * 4) * kHeadWords * sizeof(uint32_t), st));
    // Persistent grids smaller than what fits, for the two queue launches of a cache-resident scene: they run near the vector L1's access
    // rate (0.81 accesses per clock and CU, PMC), which four and a half workgroups per CU sustain as well as six, and the
    // registers left free let the shading kernels of the other frame lanes run beside them instead of waiting for a persistent grid to
    // drain (kernel timeline, profiles/r03_timeline.txt: k_combine in flight 78 % of the time for 8 % of the work).  One GPU, batches of
    // eight frames: 1.83 -> 1.76 ms per frame.  Not for the primary launch (latency-bound: 0.24 -> 0.27 ms at 75 %) and not for a
    // scene that misses L2 (1 M triangles: any-hit launch 15.2 -> 16.1 ms at 75 % -- it needs every wave to cover the fabric's latency).
    // A tile-parallel rank traces 1/n of the rays and overlaps four batches: half-size grids (one rank of 8: 0.47 -> 0.43 ms/frame).
    int gridPct = host.g.world > 1 ? 50 : (cacheResident ? 75 : 100);
    if (const char *e = getenv("RT_GRID_PCT")) gridPct = std::max(1, atoi(e));
    int gridPctPrimary = host.g.world > 1 ? gridPct : 100;
    if (const char *e = getenv("RT_GRID_PCT_PRIMARY")) gridPctPrimary = std::max(1, atoi(e));
    if ((getenv("RT_TRACE_STATS") || getenv("RT_TRACE_TIMING")) && !w->stats) { W_TRY(hipMalloc(&w->stats, 64 * sizeof(unsigned long long))); W_TRY(hipMemset(w->stats, 0, 64 * sizeof(unsigned long long))); }
    unsigned long long *S = w->stats;
    const TraceTune tune = w->tune;
    const unsigned tilesFrame = (unsigned)std::max(host.g.nLocalTiles, 0);
    const unsigned tiles = tilesFrame * (unsigned)batch;
    if (tiles == 0) return RT_OK;
    // EXPERIMENT (RT_CU_SPLIT=k): the shading kernels run on a second stream restricted to k eighths of the CUs, the traversal
    // launches -- the lane's own stream, created with the complementary mask in rt_api.hip -- on the rest.  ss == st when the experiment is off.
    hipStream_t ss = w->shadeStream ? w->shadeStream : st;
    auto hop = [&](hipStream_t from, hipStream_t to) -> hipError_t {
        if (from == to) return hipSuccess;
        hipError_t e = hipEventRecord(w->hopEv, from);
        return e != hipSuccess ? e : hipStreamWaitEvent(to, w->hopEv, 0);
    };
    W_TRY(hop(st, ss));

    rt_stage_begin(ctx, ST_PRIMARY, ss);
    hipLaunchKernelGGL(k_primary, dim3((tiles + kAppendBatch - 1) / kAppendBatch), dim3(256), 0, ss, dFrame, tg, wb);
    rt_stage_end(ctx, ST_PRIMARY, 1, ss);
    W_TRY(hop(ss, st));

    rt_stage_begin(ctx, ST_TRACE_PRIMARY, st);
    PrimarySrc ps;
    ps.fr = dFrame; ps.cand = wb.cand; ps.count = &wb.counts[0]; ps.outT = wb.primT; ps.outTri = wb.primTri;
    TraceTune tuneP = tune;   // primary rays: their cost varies strongly across the screen, so shorter runs than the queue launches (128 - 256, see k_trace)
    tuneP.chunkMax = 256;
    if (const char *e = getenv("RT_CHUNK_PRIMARY")) tuneP.chunk = atoi(e);
    launch_trace<PrimarySrc, false>(st, w->cus, gridPctPrimary, treeDepth, dFrame, host.sc, ps, &wb.heads[0], w->acc + 2, w->acc + 8, tuneP, S ? S + 0 : nullptr);
    rt_stage_end(ctx, ST_TRACE_PRIMARY, 1, st);

    W_TRY(hop(st, ss));
    rt_stage_begin(ctx, ST_POST_PRIMARY, ss);
    hipLaunchKernelGGL(k_post_primary, dim3((tiles + kAppendBatch - 1) / kAppendBatch), dim3(256), 0, ss, dFrame, tg, wb);
    rt_stage_end(ctx, ST_POST_PRIMARY, 1, ss);
    W_TRY(hop(ss, st));

    // More than one chunk: the number of chunks that hold hits is known only on the device.  Launching the chunk loop for the upper
    // bound (rounds 1-2) costs little time -- the kernels of an empty chunk return at once -- but fills the launch statistics with
    // empty launches (bench.py's batches of eight 1080p frames: 5 launch sets, 2 of them with rays).  Such a batch is tens of
    // milliseconds of work, so the hit count is read back once (this lane's stream only; the other lanes keep the GPU busy) and
    // the hits are dealt over equal chunks.  A single-chunk frame -- every frame-by-frame BASELINE configuration -- never syncs.
    if ((deferred || nChunks > 1) && !w->chunksFromSlots) {
        W_TRY(hipMemcpyAsync(w->hostHits, &w->counts[1], sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        W_TRY(hipStreamSynchronize(st));
        const size_t hits = *w->hostHits;
        if (giCopied > 0 && w->hostHits[1] > 0) {     // the previous launch set of this lane: what share of its (hit, sample) pairs had a bounce hit
            unsigned long long bounced = 0;
            for (int i = 0; i < giCopied; ++i) bounced += w->hostHits[2 + i];
            double &frac = w->pool ? w->pool->q2Frac : w->q2FracOwn;
            frac = std::max(frac, std::max(1e-9, (double)bounced / ((double)w->hostHits[1] * (double)std::max(w->prevSPP, 1))));
        }
        nChunks = (int)((hits + CHbudget - 1) / CHbudget);
        if (nChunks > 0) CH = align_up((hits + (size_t)nChunks - 1) / (size_t)nChunks, 256);
        // a set of several chunks keeps its arena for the whole chunk loop -- shared by two lanes that would serialise the lanes (4 spp -> 16 spp, 4K and
        // the 1 M-triangle scene lost 3 % that way) -- so it takes the lane's OWN arena; one-chunk sets (every frame-by-frame configuration, bench.py's
        // batches) share lane % n
        if (w->pool && deferred) w->arena = nChunks > 1 ? w->lane : w->lane % w->pool->n;   // (not deferred: chosen and allocated before k_primary)
        predictQ2 = deferred;
        if (deferred && nChunks > 0) {
            // the arenas hold this launch set's hits (+ 6 % when they have to grow); a set of several chunks gets the whole budget at once: its chunk size
            // changes with the number of frames in the batch (20 M hits = 3 x 6.7 M, 17 M = 2 x 8.6 M), and growing a 16 GB arena in the middle of a run
            // is a device-wide synchronisation plus a large allocation (the 1 M-triangle scene at --steps 20: 47.6 instead of 18.7 ms per frame)
            int rc = ensure(CH, nChunks > 1 ? CHbudget : std::min(CHbudget, align_up(CH + CH / 16, 256)));
            if (rc != RT_OK) return rc;
        }
        if (nChunks > 0) carve(CH);
    }
    // shared arena: everything from here to the last combine reads or writes it
    if (w->pool && nChunks > 0 && w->pool->lastUser[w->arena] && w->pool->lastUser[w->arena] != st) W_TRY(hipStreamWaitEvent(st, w->pool->freeEv[w->arena], 0));
    for (int c = 0; c < nChunks; ++c) {
        const uint32_t c0 = (uint32_t)((size_t)c * CH);
        const unsigned gridHS = (unsigned)((CH * (size_t)SPP + 255) / 256), gridH = (unsigned)((CH + 255) / 256);
        W_TRY(hop(st, ss));
        rt_stage_begin(ctx, ST_GEN_DIRECT, ss);
        hipLaunchKernelGGL(k_gen_direct, dim3(gridHS), dim3(256), 0, ss, dFrame, wb, c0);
        rt_stage_end(ctx, ST_GEN_DIRECT, 1, ss);
        W_TRY(hop(ss, st));

        // AO rays: one packet per hit (k_trace_packets); the any-hit launch below then starts behind the A AO slots of queue 1
        const bool pkAO = w->packetAO && A > 0;
        if (pkAO) {
            PacketSrc pk;
            pk.o = wb.shO; pk.d = wb.shD; pk.tm = wb.shT; pk.occ = wb.occ1; pk.liveCount = &wb.counts[1]; pk.c0 = c0; pk.cap = wb.CH; pk.stride = wb.CH; pk.A = A; pk.nLive = 0;
            rt_stage_begin(ctx, ST_TRACE_AO, st);
            launch_packets(st, w->cus, gridPct, treeDepth, dFrame, host.sc, pk, &wb.heads[(size_t)(1 + c * 4 + 3) * kHeadWords], w->acc + 7, w->acc + 14, tune);
            rt_stage_end(ctx, ST_TRACE_AO, 1, st);
        }
        const size_t skip = pkAO ? (size_t)A * CH : 0;
        QueueSrc q1;
        q1.o = wb.shO + skip; q1.d = wb.shD + skip; q1.tm = wb.shT + skip; q1.liveCount = &wb.counts[1]; q1.c0 = c0; q1.cap = wb.CH; q1.stride = wb.CH;
        q1.slots = (uint32_t)(S1 - (pkAO ? A : 0));
        q1.denseSlots = pkAO ? 0u : (uint32_t)A;
        q1.outT = nullptr; q1.outTri = nullptr; q1.outOcc = wb.occ1 + skip;
        if (u.enableGI == 1) {
            // bounce rays first, then ONE any-hit launch over both shadow queues
            QueueSrc qg;
            qg.o = wb.giO; qg.d = wb.giD; qg.tm = wb.giL; qg.liveCount = &wb.counts[1]; qg.c0 = c0; qg.cap = wb.CH; qg.stride = wb.CH; qg.slots = (uint32_t)SPP; qg.denseSlots = (uint32_t)SPP;
            qg.outT = wb.giT; qg.outTri = wb.giTri; qg.outOcc = nullptr;
            rt_stage_begin(ctx, ST_TRACE_GI, st);
            launch_trace<QueueSrc, false>(st, w->cus, gridPct, treeDepth, dFrame, host.sc, qg, &wb.heads[(size_t)(1 + c * 4 + 1) * kHeadWords], w->acc + 4, w->acc + 10, tune, S ? S + 32 : nullptr);
            rt_stage_end(ctx, ST_TRACE_GI, 1, st);

            W_TRY(hop(st, ss));
            rt_stage_begin(ctx, ST_GEN_GI, ss);
            hipLaunchKernelGGL(k_gen_gi, dim3(gridHS), dim3(256), 0, ss, dFrame, wb, c0, &wb.counts[64 + c]);
            if (wb.q2Stride < wb.CH * (uint32_t)SPP)   // a predicted capacity: the pairs beyond it (normally none) trace their rays in place
                hipLaunchKernelGGL(k_gen_gi_overflow, dim3(std::min<unsigned>(gridHS, (unsigned)w->cus * 4u)), dim3(256), (size_t)256 * std::max(treeDepth, 4) * 8, ss, dFrame, wb, c0, &wb.counts[64 + c], std::max(treeDepth, 4));
            rt_stage_end(ctx, ST_GEN_GI, 1, ss);
            W_TRY(hop(ss, st));

            DualQueueSrc qq;
            qq.a = q1;
            qq.b.o = wb.sh2O; qq.b.d = wb.sh2D; qq.b.tm = wb.sh2T; qq.b.liveCount = &wb.counts[64 + c]; qq.b.c0 = 0; qq.b.cap = wb.q2Stride;
            qq.b.stride = wb.q2Stride; qq.b.slots = 6u; qq.b.denseSlots = 0u;
            qq.b.outT = nullptr; qq.b.outTri = nullptr; qq.b.outOcc = wb.occ2;
            rt_stage_begin(ctx, ST_TRACE_SHADOW, st);
            launch_trace<DualQueueSrc, true>(st, w->cus, gridPct, treeDepth, dFrame, host.sc, qq, &wb.heads[(size_t)(1 + c * 4 + 0) * kHeadWords], w->acc + 3, w->acc + 9, tune, S ? S + 16 : nullptr);
            rt_stage_end(ctx, ST_TRACE_SHADOW, 1, st);
        } else {
            rt_stage_begin(ctx, ST_TRACE_SHADOW, st);
            launch_trace<QueueSrc, true>(st, w->cus, gridPct, treeDepth, dFrame, host.sc, q1, &wb.heads[(size_t)(1 + c * 4 + 0) * kHeadWords], w->acc + 3, w->acc + 9, tune, S ? S + 16 : nullptr);
            rt_stage_end(ctx, ST_TRACE_SHADOW, 1, st);
        }
        // the last traversal launch of the batch is queued: the shared ray arena may go to the next batch (k_combine reads the lane's own result arrays)
        if (w->pool && c == nChunks - 1) { W_TRY(hipEventRecord(w->pool->freeEv[w->arena], st)); w->pool->lastUser[w->arena] = st; }
        W_TRY(hop(st, ss));
        rt_stage_begin(ctx, ST_COMBINE, ss);
        hipLaunchKernelGGL(k_combine, dim3(gridH), dim3(256), 0, ss, dFrame, tg, wb, c0);
        rt_stage_end(ctx, ST_COMBINE, 1, ss);
        W_TRY(hop(ss, st));
    }
    w->prevChunks = (u.enableGI == 1) ? nChunks : 0;
    w->prevSPP = SPP;
    hipLaunchKernelGGL(k_accum_tally, dim3(1), dim3(64), 0, st, w->counts, w->acc, batch);
    // temporal resolve: the one stage that needs the previous frame's COLOR0 (and must not overtake its target stores)
    if (evPrevDone) W_TRY(hipStreamWaitEvent(st, evPrevDone, 0));
    rt_stage_begin(ctx, ST_RESOLVE, st);
    hipLaunchKernelGGL(k_resolve, dim3(tilesFrame), dim3(256), 0, st, dFrame, tg, wb);
    rt_stage_end(ctx, ST_RESOLVE, 1, st);
    W_TRY(hipGetLastError());
    return RT_OK;
}

// Closest-hit traversal of the rays listed in idx[0 .. *count) (queue addresses into o / d; results to outT / outTri at the same address) with the
// persistent kernel of this pipeline, for rt_hybrid.hip.  `heads`: kHeadWords zeroed cursor words.
void rt_wave_trace_closest_indexed(hipStream_t st, int cus, int treeDepth, const DevFrame *dFrame, const DevScene &hostScene, const uint32_t *idx,
                                   const uint32_t *count, const float4 *o, const float4 *d, float *outT, int *outTri, uint32_t *heads) {
    IndexedSrc q;
    q.idx = idx; q.count = count; q.o = o; q.d = d; q.outT = outT; q.outTri = outTri; q.n = 0;
    TraceTune tune = default_tune();
    launch_trace<IndexedSrc, false>(st, cus, 100, treeDepth, dFrame, hostScene, q, heads, nullptr, nullptr, tune, nullptr);
}
void rt_wave_trace_closest_compact(hipStream_t st, int cus, int treeDepth, const DevFrame *dFrame, const DevScene &hostScene, const float4 *o, const float4 *d,
                                   const uint32_t *dst, const uint32_t *count, const uint32_t *flags, uint32_t cap, float *outT, int *outTri, uint32_t *heads, uint32_t capOut) {
    CompactSrc q;
    q.o = o; q.d = d; q.dst = dst; q.count = count; q.flags = flags; q.cap = cap; q.capOut = capOut; q.outT = outT; q.outTri = outTri; q.n = 0;
    TraceTune tune = default_tune();
    launch_trace<CompactSrc, false>(st, cus, 100, treeDepth, dFrame, hostScene, q, heads, nullptr, nullptr, tune, nullptr);
}
// Diagnostics (rt_debug_trace kinds 2 / 3): n arbitrary rays through the PRODUCTION traversal kernels -- a one-slot queue of n entries, closest-hit or any-hit
// launch as the frames use it (persistent grid, refill scheduler, the any-hit node form rt_upload_bvh chose) -- so that the kernels can be tested ray by ray.
void rt_wave_debug_trace(hipStream_t st, int cus, int treeDepth, const DevFrame *dFrame, const DevScene &hostScene, bool any, const float4 *o, const float4 *d,
                         const float *tm, const uint32_t *liveCount, uint32_t n, float *outT, int *outTri, uint8_t *outOcc, uint32_t *heads) {
    QueueSrc q;
    q.o = o; q.d = d; q.tm = tm; q.liveCount = liveCount; q.c0 = 0; q.cap = n; q.stride = n; q.slots = 1; q.denseSlots = 0;
    q.outT = outT; q.outTri = outTri; q.outOcc = outOcc; q.nLive = 0;
    TraceTune tune = default_tune();
    if (const char *e = getenv("RT_QNODES")) tune.qnodes = atoi(e);
    if (any) launch_trace<QueueSrc, true>(st, cus, 100, treeDepth, dFrame, hostScene, q, heads, nullptr, nullptr, tune, nullptr);
    else launch_trace<QueueSrc, false>(st, cus, 100, treeDepth, dFrame, hostScene, q, heads, nullptr, nullptr, tune, nullptr);
}
size_t rt_wave_head_words() { return kHeadWords; }

int rt_wave_traced(RtWave *w, hipStream_t st, unsigned long long *out8, bool reset) {   // 16 words, see rt_wave.hpp
    for (int i = 0; i < 16; ++i) out8[i] = 0;
    if (!w->acc) return RT_OK;
    W_TRY(hipStreamSynchronize(st));
    W_TRY(hipMemcpy(out8, w->acc, 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    if (w->stats) {
        unsigned long long v[64];
        W_TRY(hipMemcpy(v, w->stats, sizeof v, hipMemcpyDeviceToHost));
        static const char *nm[4] = {"primary", "shadow", "bounce", "bounce_shadow"};
        if (getenv("RT_TRACE_TIMING") && atoi(getenv("RT_TRACE_TIMING"))) {
            // RT_TRACE_TIMING=1 (k_trace<.., TIMING>): sums over all waves of the launches since the last reset
            for (int k = 0; k < 3; ++k) {
                const unsigned long long *q = v + k * 16;
                if (!q[0] && !q[4]) continue;
                const double it = (double)std::max<unsigned long long>(q[0], 1), lf = (double)std::max<unsigned long long>(q[4], 1), rf = (double)std::max<unsigned long long>(q[6], 1);
                fprintf(stderr, "[trace timing] %-8s wave lifetime %.4g cycles | inner iterations %.4g: to load issue %.0f + loads in flight %.0f + after arrival %.0f cycles each (%.2f / %.2f / %.2f of the "
                                "lifetime) | leaf phases %.4g: %.0f cycles each (%.2f) | refill rounds %.4g: %.0f cycles each (%.2f)\n",
                        nm[k], (double)q[8], it, q[1] / it, q[2] / it, q[3] / it, (double)q[1] / (double)std::max<unsigned long long>(q[8], 1), (double)q[2] / (double)std::max<unsigned long long>(q[8], 1),
                        (double)q[3] / (double)std::max<unsigned long long>(q[8], 1), lf, q[5] / lf, (double)q[5] / (double)std::max<unsigned long long>(q[8], 1), rf, q[7] / rf,
                        (double)q[7] / (double)std::max<unsigned long long>(q[8], 1));
            }
            if (reset) W_TRY(hipMemset(w->stats, 0, sizeof v));
            if (reset) W_TRY(hipMemset(w->acc, 0, 16 * sizeof(unsigned long long)));
            return RT_OK;
        }
        const char *mode = getenv("RT_TRACE_STATS");
        const bool quiet = mode && atoi(mode) >= 2;   // RT_TRACE_STATS=2: collect (rt_get_traced_rays' mergedLoads*), do not print
        for (int k = 0; k < 4; ++k) {
            const unsigned long long *q = v + k * 16;
            double rays = (double)std::max<unsigned long long>(out8[2 + k], 1);
            if (!quiet) fprintf(stderr, "[trace stats] %-13s rays %.3g | per ray: inner %.1f leaf %.1f tri %.1f | inner-phase lane util %.2f (%.3g wave-iters, %.0f cyc each) "
                            "leaf-phase util %.2f (%.3g, %.0f cyc each) | refills %.3g (%.0f cyc each) | gather loads per ray %.0f, merged within lane quads %.0f (nodes %.0f + triangles %.0f), merged wave-wide %.0f (%.0f + %.0f)\n",
                    nm[k], rays, q[0] / rays, q[1] / rays, q[2] / rays, q[3] ? q[4] / (64.0 * q[3]) : 0.0, (double)q[3], q[3] ? (double)q[8] / q[3] : 0.0,
                    q[5] ? q[6] / (64.0 * q[5]) : 0.0, (double)q[5], q[5] ? (double)q[9] / q[5] : 0.0, (double)q[7], q[7] ? (double)q[10] / q[7] : 0.0,
                    k < 3 ? out8[8 + k] / rays : 0.0, (q[12] + q[13]) / rays, q[12] / rays, q[13] / rays, (q[14] + q[15]) / rays, q[14] / rays, q[15] / rays);
            if (k < 3) out8[11 + k] = q[12] + q[13];   // k: 0 primary, 1 shadow (+ bounce-shadow: one launch), 2 bounce
        }
        if (reset) W_TRY(hipMemset(w->stats, 0, sizeof v));
    }
    if (reset) W_TRY(hipMemset(w->acc, 0, 16 * sizeof(unsigned long long)));
    return RT_OK;
}

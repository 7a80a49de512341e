// rt_hybrid.hip -- EXTENSION beyond the reference (BASELINE configs[2-3] "run B"): the hybrid scene -- the analytic
// branch of rt.frag (:108-163) with the uploaded mesh as one more object, N diffuse bounces -- rendered in STAGES instead of by the
// one-thread-per-pixel megakernel.
//
// The analytic shading code (rt_device_analytic.hpp) is a web of scene queries whose rays depend on earlier hits (glass: three
// secondary rays, each shaded with six visibility queries; mirror; N bounces), so it is not cut into stages by hand.  Instead it is
// REPLAYED: thread = (pixel, sample) runs the unchanged shading code with a Replay state (rt_device_shade.hpp) behind the one function
// all mesh queries go through (traceScene).  Queries are numbered in program order; a query answered by an earlier pass is read from
// the thread's log, an open one is written to the thread's slot of a ray queue and answered "no mesh hit" ON SPECULATION: the thread goes on
// -- recording every later query too -- as if that were the answer, and keeps the radiance it arrives at.  The next pass first checks the
// traced answers of what the thread recorded: if none of them is a mesh hit in front of the analytic scene's own hit, everything the thread
// did was right and it is finished without being shaded again (with the reference's default camera the mesh stands between the floor and the
// point light: nearly every sample has such a query, and about half of them miss the mesh); otherwise the answers up to the first such
// hit of a query that later rays were built from (primary, bounce, reflection, refraction -- visibility hits leave the later queries valid) join the
// log and the thread is shaded again from there -- in a pass over the DENSE list of such threads (k_hybrid_verify packs it), so that later passes
// run whole waves of work instead of a few lanes per wave.  At the end of a
// pass every wave appends the queue addresses of what its threads recorded to a dense list (one atomic per wave; query-major inside the
// wave's block, so that neighbouring list entries are the same kind of ray of neighbouring pixels), and between two shading passes ONE
// persistent closest-hit launch of the wavefront pipeline (k_trace over that list, rt_wave.hip) traces everything recorded.  A finished
// thread's radiance is bit-identical to the megakernel's, because every operation it executed saw the same inputs.  Passes per chunk = the
// longest chain of mesh HITS a sample's rays depend on + 2.  Rays that miss the mesh's root box never enter the queue (the test bvh_closest starts with).
#include <algorithm>
#include <cstdio>
#include <string>

#include "../../include/rt_mi355.h"
// Round 5: the shading code inlined into this kernel (rt_device_analytic.hpp "Inlining policy").  As calls -- the megakernel's form -- the Replay state and every Hit
// lived in scratch memory and each of a sample's ~40 scene queries saved and restored registers there: 1.8e10 scratch instructions and 3.6 TB/s of fabric traffic
// in a kernel that used 18 % of its vector issue slots (profiles/r05_hybrid_pmc.txt).
#ifndef RT_HYBRID_CALLS
#define RT_ANALYTIC_LEAF __forceinline__
#define RT_ANALYTIC_MID __forceinline__
#endif
#include "rt_device_analytic.hpp"
#include "rt_wave.hpp"

#pragma clang fp contract(off)

using namespace rtd;

namespace {

constexpr uint32_t kDone = 0xffffffffu;
constexpr uint32_t kCnt = 32;   // words between two counters: each on its own 128-byte line (they are hit by one atomic per workgroup)
enum { ST_TRACE_GI = 6, ST_RESOLVE = 8, ST_COMBINE = 9 };   // stage ids shared with rt_wave.hip (rt_stage_name)
// counters, at [k * kCnt]
enum { C_OPEN = 0,      // threads that recorded in the shading pass just run (= entries of recd)
       C_REC = 1,       // queries recorded in it (= entries of the dense queue; keeps counting past the capacity)
       C_FLAGS = 2,     // 1: a sample needs more than qmax queries, 2: dense queue too small, 4: log arena too small; RT_HYBRID_CHECK=1: 8 thread list,
                        // 16 dense queue, 32 log arena, 64 staging area -- a store whose index lies outside the array it was given (never seen; see hb_ok)
       C_TODO = 3,      // entries of todo (threads the next pass shades again)
       C_LOG = 4,       // bump pointer of the log arena (entries; never reset inside a chunk; keeps counting past the capacity)
       C_MAXREC = 5,    // largest C_REC of any pass of the chunk (what the dense queue has to hold)
       C_PASSES = 6,    // shading passes that had something to shade
       C_TILE = 7,      // tile cursor of the persistent shading kernel (reset before every pass)
       C_WORDS = 8 };

// Round 4: memory follows the queries that are really recorded, and the host is out of the pass loop.
//   * an open query is written to the WORKGROUP's staging area ([q][256 threads], one area per resident workgroup of the persistent shading kernel);
//     at the end of a tile of 256 threads the workgroup knows what each thread recorded, reserves -- one atomic each per workgroup -- a block of the
//     dense per-pass QUEUE (records in traversal order: query-major inside a wave's block, so that neighbouring entries are the same kind of ray of
//     neighbouring pixels) and a block of the LOG arena (per thread: its answered queries so far + room for the answers to what it just recorded,
//     contiguous), copies the thread's old log into its new block and the staged records into the queue, each with the log address its answer goes to;
//   * the traversal launch (k_trace<CompactSrc>) reads the queue densely and writes (t, triangle) straight into the asking thread's log;
//   * the check reads the thread's answers from its log, next to the analytic scene's own hit distance kept there (sign bit: later rays were built from this hit).
// Rounds 3's [qmax x threads] queue and log (80 GB for one 1080p / 16 spp / 4-bounce frame) are gone: per thread 32 bytes of state and radiance, per recorded
// query 36 bytes of queue and, per pass it survives, 12 bytes of log.  Capacities are estimates (kept across frames); a pass that outgrows them only counts,
// writes nothing, raises a flag -- the host, which looks at the counters once per block of passes, enlarges the arrays and renders the chunk again.
struct HybridBuf {
    float4 *stgO, *stgD;   // [resident workgroups][qmax][256] staging
    float4 *qO, *qD;       // [capQ] dense queue of the pass
    uint32_t *qDst;        // [capQ] log entry the answer of a record goes to
    float *logT;           // [capL] log arena: t (inf on a miss) ...
    int *logTri;           //        ... triangle (-1 on a miss) ...
    float *logLim;         //        ... and, for recorded queries, the analytic scene's own hit distance (uINF: none), negated when later rays are built from this hit
    uint32_t *state;       // [T] answered queries | recorded-up-to << 16, or kDone
    uint32_t *logBase;     // [T] the thread's block of the log arena
    float4 *rad;           // [T] final radiance of (pixel, sample)
    float2 *sMotion;       // [nS] from the sample-0 thread: rt.frag:94-101, 172-175
    float4 *sPos, *sNrm;   // [nS]
    uint32_t *todo, *recd; // [T] dense lists of threads: to shade in this pass (behind the first pass) / that recorded queries in this pass
    uint32_t *cnt;
    uint32_t slot0, nS, T, qmax, capQ, capL;
    int SPP;
    int check;             // RT_HYBRID_CHECK=1: every staged-record, thread-list, queue and log store compares its index with the capacity of its array first
};

// RT_HYBRID_CHECK=1 (diagnostic): the capacities of the dense queue and the log arena are ESTIMATES, and a pass that outgrows them is
// supposed to count only (C_FLAGS 2 / 4) and store nothing.  With the check on, every store of the staged pipeline first proves that: an index at or beyond
// the capacity its array was allocated with raises `bit` in C_FLAGS and the store is skipped; the host then ends the frame with RT_ERR_STATE.
RT_DEV bool hb_ok(const HybridBuf &hb, uint32_t idx, uint32_t cap, uint32_t bit) {
    if (!hb.check || idx < cap) return true;
    atomicOr(&hb.cnt[C_FLAGS * kCnt], bit);
    return false;
}

RT_DEV uint32_t wave_excl_scan(uint32_t v, uint32_t lane, uint32_t &total) {
    uint32_t incl = v;
    for (uint32_t off = 1; off < 64u; off <<= 1) {
        const uint32_t t = (uint32_t)__shfl_up((int)incl, off, 64);
        if (lane >= off) incl += t;
    }
    total = (uint32_t)__shfl((int)incl, 63, 64);
    return incl - v;
}

// Four waves per SIMD: 128 VGPRs and 160 bytes of scratch per lane, with the shading code INLINED (round 5; as calls it was 96 VGPRs + 800 B at five waves, and the
// scratch traffic of ~40 calls per sample was what the kernel waited on: run B 30.0 -> 17.8 ms per frame).  Measured on MI355X (1080p, 16 spp, 4 bounces, whole
// frame): 18.7 / 17.8 / 18.7 / 19.9 ms at 3 / 4 / 5 / 6 waves per SIMD (72 / 160 / 328 / 400 B of scratch); round 3, as calls, shading passes only: 41.1 / 38.0 /
// 36.4 / 39.1 ms at 3 / 4 / 5 / 6.
// Persistent: a fixed grid walks the tiles of 256 threads (first pass of a chunk: thread = (sample, slot) in order; later passes: the dense list of threads
// whose speculation failed), so that the list length can stay on the device and every workgroup owns one staging area.
template <int WAVES>
__global__ __launch_bounds__(256, WAVES) void k_hybrid_shade(const DevFrame *__restrict__ fr, HybridBuf hb, int listed) {
    __shared__ uint32_t sOpen[4], sRec[4], sLog[4], sBase[3], sOk;
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    __shared__ uint32_t sTile;
    // Once a pass of this chunk has outgrown the queue or the log arena, what it left behind is incomplete (the workgroups that did not fit wrote
    // nothing): every later kernel of the chunk returns at once, the host enlarges the arrays and renders the chunk again.
    if (listed && (hb.cnt[C_FLAGS * kCnt] & 6u)) return;
    const uint32_t n = listed ? hb.cnt[C_TODO * kCnt] : hb.T;
    const uint32_t nTiles = (n + 255u) / 256u;
    float4 *stgO = hb.stgO + (size_t)blockIdx.x * 256u * hb.qmax, *stgD = hb.stgD + (size_t)blockIdx.x * 256u * hb.qmax;
    // tiles are dealt by a cursor, one atomic per tile of 256 threads (a tile costs tens of microseconds and up; a sky tile and a tile on the glass
    // sphere differ by an order of magnitude, so a static deal leaves the slowest workgroup a tenth of the pass behind the others)
    for (;;) {
        if (threadIdx.x == 0u) sTile = atomicAdd(&hb.cnt[C_TILE * kCnt], 1u);
        __syncthreads();
        const uint32_t tile = sTile;
        if (tile >= nTiles) break;
        const uint32_t gid = tile * 256u + threadIdx.x;
        const uint32_t tid = gid < n ? (listed ? hb.todo[gid] : gid) : 0u;
        bool open = false;
        uint32_t recorded = 0, known = 0;
        if (gid < n) {
            const uint32_t st = hb.state[tid];
            if (st != kDone) {
                const RtUniforms &u = fr->u;
                const int s = (int)(tid / hb.nS);
                const uint32_t i = tid % hb.nS, slot = hb.slot0 + i;
                int px, py;
                if (!pixel_of_slot(fr->g, (int)(slot >> 8), (int)(slot & 255u), px, py)) {
                    hb.rad[tid] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);   // padding of a ragged tile: nothing to render
                    hb.state[tid] = kDone;
                } else {
                    const uint32_t kn = st & 0xffffu;          // answers [0, kn) are in the log
                    Replay R;
                    R.known = kn;
                    R.slot = threadIdx.x; R.qmax = hb.qmax;
                    R.o = stgO; R.d = stgD;
                    const uint32_t lb = kn ? hb.logBase[tid] : 0u;
                    R.logT = hb.logT + lb; R.logTri = hb.logTri + lb;
                    Frag F;
                    F.u = &u; F.sc = &fr->sc; F.fcx = (float)px + 0.5f; F.fcy = (float)py + 0.5f;
                    F.stk = nullptr; F.rp = &R; F.giBounces = fr->giBounces; F.frameIndex = u.frameIndex;
                    Work w;
                    work_zero(w);
                    const V3 camPos = ld3(u.camPos);
                    const V3 dir = primaryDir(u, F.fcx, F.fcy);
                    const int SPP = max(u.spp, 1);
                    const int seed = (int)((uint32_t)u.frameIndex * (uint32_t)SPP + (uint32_t)s);
                    // one sample of rt.frag:118-176, as in k_mega's analytic branch
                    Hit h;
                    const bool hitAny = traceScene<false>(F, camPos, dir, true, true, h, w, true);
                    V3 radiance;
                    V2 motion = mk2(0.0f, 0.0f);
                    V4 gpos = mk4(0.0f, 0.0f, 0.0f, 0.0f), gnrm = mk4(0.0f, 0.0f, 0.0f, 0.0f);
                    if (hitAny) {
                        if (s == 0) {
                            V2 prevNDC = ndcFromWorld(h.p, u.prevViewProj), currNDC = ndcFromWorld(h.p, u.currViewProj);
                            motion = mk2(currNDC.x - prevNDC.x, currNDC.y - prevNDC.y);
                            gpos = mk4(h.p.x, h.p.y, h.p.z, 1.0f);
                            V3 nn = normalize(h.n);
                            gnrm = mk4(nn.x, nn.y, nn.z, 0.0f);
                        }
                        radiance = shadeSampleAnalytic<false>(F, h, -dir, seed, w);
                    } else {
                        radiance = sky<false>(F, dir, w);
                        if (u.cameraMoved == 1 && s == 0) motion = mk2(4.0f, 4.0f);
                    }
                    if (R.overflow) atomicOr(&hb.cnt[C_FLAGS * kCnt], 1u);
                    // the result of this pass is kept either way: final if nothing was open, else provisional until the next pass has checked the speculation
                    hb.rad[tid] = make_float4(radiance.x, radiance.y, radiance.z, 0.0f);
                    if (s == 0) {
                        hb.sMotion[i] = make_float2(motion.x, motion.y);
                        hb.sPos[i] = make_float4(gpos.x, gpos.y, gpos.z, gpos.w);
                        hb.sNrm[i] = make_float4(gnrm.x, gnrm.y, gnrm.z, gnrm.w);
                    }
                    if (R.pending == 0u) {
                        hb.state[tid] = kDone;
                    } else {
                        open = true;
                        known = R.known;
                        const uint32_t recEnd = max(R.recEnd, R.known);
                        recorded = recEnd - R.known;   // queries [known, recEnd) were recorded: traced by the next launch
                        hb.state[tid] = R.known | (recEnd << 16);
                    }
                }
            }
        }
        // ---- pack what this tile recorded.  ONE atomic per counter and WORKGROUP: a single counter word takes about 90 M atomics per second on MI355X,
        // and one pair per wave (224 k waves in the first pass of a 1080p / 16 spp chunk) was most of that pass's time in round 3's first version.
        const unsigned long long om = __ballot(open);
        const uint32_t logLen = open ? known + recorded : 0u;
        uint32_t wRec, wLog;
        const uint32_t lofs = wave_excl_scan(logLen, lane, wLog);
        (void)wave_excl_scan(recorded, lane, wRec);
        uint32_t most = recorded;
        for (int off = 32; off > 0; off >>= 1) most = max(most, (uint32_t)__shfl_down((int)most, off, 64));
        most = (uint32_t)__shfl((int)most, 0, 64);
        if (lane == 0u) { sOpen[wv] = (uint32_t)__popcll(om); sRec[wv] = wRec; sLog[wv] = wLog; }
        __syncthreads();
        if (threadIdx.x == 0u) {
            const uint32_t nOpen = sOpen[0] + sOpen[1] + sOpen[2] + sOpen[3], nRec = sRec[0] + sRec[1] + sRec[2] + sRec[3], nLog = sLog[0] + sLog[1] + sLog[2] + sLog[3];
            sBase[0] = nOpen ? atomicAdd(&hb.cnt[C_OPEN * kCnt], nOpen) : 0u;
            sBase[1] = nRec ? atomicAdd(&hb.cnt[C_REC * kCnt], nRec) : 0u;
            sBase[2] = nLog ? atomicAdd(&hb.cnt[C_LOG * kCnt], nLog) : 0u;
            uint32_t flags = 0u;
            if ((unsigned long long)sBase[1] + nRec > hb.capQ) flags |= 2u;
            if ((unsigned long long)sBase[2] + nLog > hb.capL) flags |= 4u;
            if (flags) atomicOr(&hb.cnt[C_FLAGS * kCnt], flags);
            sOk = flags == 0u;
        }
        __syncthreads();
        if (sOk && om != 0ull) {
            uint32_t tbase = sBase[0], qbase = sBase[1], lbase = sBase[2];
            for (uint32_t k = 0; k < wv; ++k) { tbase += sOpen[k]; qbase += sRec[k]; lbase += sLog[k]; }
            const unsigned long long lt = (1ull << lane) - 1ull;
            const uint32_t newLog = lbase + lofs;
            if (open) {
                const uint32_t rpos = tbase + (uint32_t)__popcll(om & lt);
                if (hb_ok(hb, rpos, hb.T, 8u)) hb.recd[rpos] = tid;      // whose speculation the next pass checks
                if (known) {
                    const uint32_t old = hb.logBase[tid];
                    for (uint32_t q = 0; q < known; ++q)
                        if (hb_ok(hb, newLog + q, hb.capL, 32u) && hb_ok(hb, old + q, hb.capL, 32u)) { hb.logT[newLog + q] = hb.logT[old + q]; hb.logTri[newLog + q] = hb.logTri[old + q]; }
                }
                hb.logBase[tid] = newLog;
            }
            for (uint32_t k = 0; k < most; ++k) {
                const unsigned long long m = __ballot(recorded > k);
                if (recorded > k) {
                    const uint32_t pos = qbase + (uint32_t)__popcll(m & lt);
                    const size_t src = (size_t)(known + k) * 256u + threadIdx.x;
                    const uint32_t a = newLog + known + k;
                    if (hb_ok(hb, known + k, hb.qmax, 64u) && hb_ok(hb, pos, hb.capQ, 16u) && hb_ok(hb, a, hb.capL, 32u)) {
                        const float4 oo = stgO[src], dd = stgD[src];
                        hb.qO[pos] = oo; hb.qD[pos] = dd; hb.qDst[pos] = a;
                        hb.logLim[a] = dd.w != 0.0f ? -oo.w : oo.w;
                    }
                }
                qbase += (uint32_t)__popcll(m);
            }
        }
        __syncthreads();   // the staging area and the shared words are reused by the next tile
    }
}

// after a shading pass: the largest queue any pass of the chunk asked for, and how many passes had work (one thread)
__global__ void k_hybrid_note(uint32_t *cnt) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        cnt[C_MAXREC * kCnt] = max(cnt[C_MAXREC * kCnt], cnt[C_REC * kCnt]);
        if (cnt[C_OPEN * kCnt] != 0u || cnt[C_REC * kCnt] != 0u) cnt[C_PASSES * kCnt]++;
    }
}

// Between a traversal launch and the next shading pass: every thread that recorded queries checks its speculation against the traced answers.
// All of them misses (or hits behind the analytic scene's own): the thread is finished, its stored radiance final.  Otherwise the answers up to the
// first failed query that later rays were built from join the log, and the thread goes onto the dense list of the next pass.
__global__ __launch_bounds__(256) void k_hybrid_verify(HybridBuf hb) {
    __shared__ uint32_t sAgain[4], sBase;
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    if (hb.cnt[C_FLAGS * kCnt] & 6u) return;   // an overflowed pass: its lists are incomplete (see k_hybrid_shade)
    const uint32_t n = hb.cnt[C_OPEN * kCnt];
    const uint32_t nTiles = (n + 255u) / 256u;
    for (uint32_t tile = blockIdx.x; tile < nTiles; tile += gridDim.x) {
        const uint32_t j = tile * 256u + threadIdx.x;
        bool again = false;
        uint32_t tid = 0;
        if (j < n) {
            tid = hb.recd[j];
            if (!hb_ok(hb, tid, hb.T, 8u)) tid = 0u;
            const uint32_t st = hb.state[tid];
            const uint32_t kn = st & 0xffffu, rec = st >> 16;
            const uint32_t base = hb.logBase[tid];
            bool anyHit = false;
            uint32_t valid = rec;                      // answers [kn, valid) are answers to queries the true frame asks
            for (uint32_t q = kn; q < rec; ++q) {
                const uint32_t a = base + q;
                if (!hb_ok(hb, a, hb.capL, 32u)) break;
                const float lim = hb.logLim[a];
                if (hb.logTri[a] >= 0 && hb.logT[a] < __builtin_fabsf(lim)) {   // a mesh hit in front of the analytic scene's: the speculation failed here
                    anyHit = true;
                    if (lim < 0.0f) { valid = q + 1u; break; }                   // later rays were built from the wrong hit: what follows is void
                }
            }
            if (!anyHit) hb.state[tid] = kDone;        // every speculated miss was one
            else { hb.state[tid] = valid; again = true; }
        }
        const unsigned long long am = __ballot(again);
        if (lane == 0u) sAgain[wv] = (uint32_t)__popcll(am);
        __syncthreads();
        if (threadIdx.x == 0u) {
            const uint32_t nAgain = sAgain[0] + sAgain[1] + sAgain[2] + sAgain[3];
            sBase = nAgain ? atomicAdd(&hb.cnt[C_TODO * kCnt], nAgain) : 0u;      // one atomic per workgroup (see k_hybrid_shade)
        }
        __syncthreads();
        if (again) {
            uint32_t base = sBase;
            for (uint32_t k = 0; k < wv; ++k) base += sAgain[k];
            const uint32_t tpos = base + (uint32_t)__popcll(am & ((1ull << lane) - 1ull));
            if (hb_ok(hb, tpos, hb.T, 8u)) hb.todo[tpos] = tid;
        }
        __syncthreads();
    }
}

// thread = pixel: the sample sum in the shader's order (rt.frag:79-184), TAA resolve, four target stores -- the tail of k_mega
__global__ __launch_bounds__(256) void k_hybrid_resolve(const DevFrame *__restrict__ fr, Targets tg, HybridBuf hb) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= hb.nS) return;
    const RtUniforms &u = fr->u;
    const uint32_t slot = hb.slot0 + i;
    int px, py;
    if (!pixel_of_slot(fr->g, (int)(slot >> 8), (int)(slot & 255u), px, py)) return;
    const int SPP = max(u.spp, 1);
    V3 frameSum = mk3(0.0f);
    for (int s = 0; s < SPP; ++s) {
        const float4 r = hb.rad[(size_t)s * hb.nS + i];
        frameSum = frameSum + mk3(r.x, r.y, r.z);
    }
    const V3 curr = frameSum / (float)SPP;
    const float2 m = hb.sMotion[i];
    const V2 motionOut = mk2(m.x, m.y);
    const float uvx = ((float)px + 0.5f) / (float)fr->g.W, uvy = ((float)py + 0.5f) / (float)fr->g.H;
    const V2 taaMotion = (u.cameraMoved == 1) ? motionOut : mk2(0.0f, 0.0f);
    HistoryTex hist;
    hist.prev = tg.prev; hist.prevAll = tg.prevAll; hist.blockSlots = tg.blockSlots; hist.g = &fr->g; hist.slot = (int)slot;
    const V4 taa = resolveTAA(u, curr, uvx, uvy, taaMotion, u.frameIndex, hist);
    const float4 gp = hb.sPos[i], gn = hb.sNrm[i];
    tg.color[slot] = pack_half4(taa);
    tg.motion[slot] = pack_half2(motionOut);
    tg.gpos[slot] = pack_half4(mk4(gp.x, gp.y, gp.z, gp.w));
    tg.gnrm[slot] = pack_half4(mk4(gn.x, gn.y, gn.z, gn.w));
}

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

}  // namespace

struct RtHybrid {
    std::string err;
    int cus = 256;
    // Budget of the per-thread state, the dense queue and the log arena together (the staging areas -- one per resident workgroup of the shading kernel,
    // qmax x 256 x 32 B each: 566 MB at four bounces -- come on top).  A 1080p / 16 spp / 4-bounce frame needs about 3.5 GB, 4K about 14 GB: one chunk each.
    size_t budgetBytes = (size_t)24 << 30;
    // expected entries per thread of the dense queue (largest pass) and of the log arena (all passes of a chunk): start values, raised when a chunk outgrows
    // them (it is then rendered again) and kept for the frames that follow
    double ratioQ = 1.0, ratioL = 2.5;
    hipEvent_t evFree = nullptr;          // recorded after a frame's last kernel: the next frame (another lane's stream) waits for it before it touches the arena
    void *arena = nullptr, *staging = nullptr;
    size_t arenaBytes = 0, stagingBytes = 0;
    int gridShade = 0;
    int waves = 4;            // launch bound of the shading kernel (waves per SIMD), RT_HYBRID_WAVES
    uint32_t *cnt = nullptr, *heads = nullptr;
    uint32_t *hostCnt = nullptr;       // pinned
    unsigned long long passes = 0, launches = 0, redone = 0;
    bool debug = false;                // RT_HYBRID_DEBUG=1: per-pass counts on stderr (one host round trip per pass)
    bool check = false;                // RT_HYBRID_CHECK=1: bounds-checked stores (hb_ok)
    size_t okSlots = 0;                // largest chunk of pixel slots the arena could be allocated for after an out-of-memory answer (0: none seen); later chunks and frames start from it
};

RtHybrid *rt_hybrid_create(int cus) {
    RtHybrid *h = new RtHybrid();
    h->cus = cus > 0 ? cus : 256;
    if (const char *e = getenv("RT_QUEUE_BUDGET_MB")) h->budgetBytes = (size_t)atoll(e) << 20;
    if (const char *e = getenv("RT_HYBRID_RATIO_Q")) h->ratioQ = std::max(0.001, atof(e));   // tests: start so small that the overflow path runs
    if (const char *e = getenv("RT_HYBRID_RATIO_L")) h->ratioL = std::max(0.001, atof(e));
    (void)hipEventCreateWithFlags(&h->evFree, hipEventDisableTiming);
    if (const char *e = getenv("RT_HYBRID_DEBUG")) h->debug = atoi(e) != 0;
    if (const char *e = getenv("RT_HYBRID_CHECK")) h->check = atoi(e) != 0;
    return h;
}
void rt_hybrid_destroy(RtHybrid *h) {
    if (!h) return;
    if (h->arena) (void)hipFree(h->arena);
    if (h->staging) (void)hipFree(h->staging);
    if (h->cnt) (void)hipFree(h->cnt);
    if (h->heads) (void)hipFree(h->heads);
    if (h->hostCnt) (void)hipHostFree(h->hostCnt);
    if (h->evFree) (void)hipEventDestroy(h->evFree);
    delete h;
}
const char *rt_hybrid_error(const RtHybrid *h) { return h->err.c_str(); }
size_t rt_hybrid_arena_bytes(const RtHybrid *h) { return h ? h->arenaBytes + h->stagingBytes : 0; }

#define H_TRY(expr)                                                                   \
    do {                                                                              \
        hipError_t e_ = (expr);                                                       \
        if (e_ != hipSuccess) { h->err = std::string(#expr) + ": " + hipGetErrorString(e_); return RT_ERR_HIP; } \
    } while (0)

int rt_hybrid_render(RtHybrid *h, RtContext *ctx, hipStream_t st, const DevFrame *dFrame, const DevFrame &host, Targets tg, int treeDepth,
                     hipEvent_t evPrevDone) {
    const RtUniforms &u = host.u;
    const size_t nSlots = (size_t)std::max(host.g.nLocalTiles, 0) * 256;
    if (nSlots == 0) return RT_OK;
    const int SPP = std::max(u.spp, 1);
    // mesh queries of one sample, worst case (every ray meets the mesh's root box): primary 1, direct 6, AO aoSamples, per bounce 1 + 6;
    // mirror: 1 + 6 more in front of its bounces; glass: 3 x (1 + 6).  A thread that needs more ends the frame with RT_ERR_UNSUPPORTED.
    const uint32_t qmax = (uint32_t)(1 + 6 + std::max(u.aoSamples, 0) + 7 * std::max(host.giBounces, 1) + 7 + 8);
    if (qmax > 0xfff0u) { h->err = "hybrid: query log too long"; return RT_ERR_UNSUPPORTED; }
    if (!h->cnt) {
        H_TRY(hipMalloc((void **)&h->cnt, C_WORDS * kCnt * sizeof(uint32_t)));
        H_TRY(hipMalloc((void **)&h->heads, rt_wave_head_words() * sizeof(uint32_t)));
        H_TRY(hipHostMalloc((void **)&h->hostCnt, C_WORDS * kCnt * sizeof(uint32_t)));
    }
    if (h->gridShade == 0) {
        int perCU = 0;
        h->waves = 4;
        if (const char *e = getenv("RT_HYBRID_WAVES")) h->waves = atoi(e) == 5 ? 5 : atoi(e) == 6 ? 6 : atoi(e) == 3 ? 3 : 4;   // EXPERIMENT: register budget of the shading kernel
        auto kfn = h->waves == 3 ? k_hybrid_shade<3> : h->waves == 5 ? k_hybrid_shade<5> : h->waves == 6 ? k_hybrid_shade<6> : k_hybrid_shade<4>;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCU, kfn, 256, 0) != hipSuccess || perCU < 1) perCU = 1;
        h->gridShade = h->cus * std::min(perCU, 8);
    }
    // staging: one [qmax][256] area of origins and one of directions per resident workgroup
    const size_t stgEach = (size_t)h->gridShade * 256 * qmax * 16;
    if (h->stagingBytes < 2 * stgEach) {
        if (h->staging) { H_TRY(hipDeviceSynchronize()); (void)hipFree(h->staging); }
        h->staging = nullptr; h->stagingBytes = 0;
        H_TRY(hipMalloc(&h->staging, 2 * stgEach));
        h->stagingBytes = 2 * stgEach;
    }
    // every exit below this point -- errors included -- leaves the event behind that the next frame (on another lane's stream) waits for before
    // it touches the shared arena and counters (an error return used to skip it)
    struct FreeGuard {
        RtHybrid *h; hipStream_t st;
        ~FreeGuard() { if (h->evFree) (void)hipEventRecord(h->evFree, st); }
    } guard{h, st};
    if (h->evFree) H_TRY(hipStreamWaitEvent(st, h->evFree, 0));   // the previous frame's kernels (on another lane's stream) still use the arena
    bool waited = false;

    auto bytes_for = [&](size_t nS, size_t capQ, size_t capL) {
        const size_t T = nS * (size_t)SPP;
        return align_up(capQ * 16, 256) * 2 + align_up(capQ * 4, 256) + align_up(capL * 4, 256) * 3 + align_up(T * 4, 256) * 4 + align_up(T * 16, 256) +
               align_up(nS * 8, 256) + align_up(nS * 16, 256) * 2 + 4096;
    };
    size_t slot0 = 0;
    int attempts = 0;
    while (slot0 < nSlots) {
        // chunk of pixel slots from the budget, at the current capacity estimates
        const double perSlot = (double)SPP * (32.0 + h->ratioQ * 36.0 + h->ratioL * 12.0) + 40.0;
        size_t nS = std::min(nSlots - slot0, std::max<size_t>((size_t)((double)h->budgetBytes / perSlot), 256));
        if (h->okSlots) nS = std::min(nS, h->okSlots);      // what the device could give last time it refused the budget-sized arena
        nS = std::max<size_t>(nS / 256 * 256, 256);
        nS = std::min(nS, nSlots - slot0);      // nSlots is a multiple of 256
        size_t T = nS * (size_t)SPP;
        while (T >= ((size_t)1 << 32) / 4) { nS = std::max<size_t>(nS / 2 / 256 * 256, 256); T = nS * (size_t)SPP; }   // 32-bit thread ids and log addresses
        size_t capQ = std::max<size_t>((size_t)((double)T * h->ratioQ) + 4096, 4096), capL = std::max<size_t>((size_t)((double)T * h->ratioL) + 4096, 4096);
        capQ = std::min(capQ, T * (size_t)qmax);
        capL = std::min<size_t>(std::min(capL, T * (size_t)qmax * 4), 0xfffffff0u);
        size_t need = bytes_for(nS, capQ, capL);
        if (h->arenaBytes < need) {
            // the arena grows on demand; when the device cannot give that much (other contexts, the wavefront pipeline's arenas), the chunk is halved
            // until it fits -- the chunked path is the tested one
            // (freeing after a sync of `st` alone is enough: `st` has waited on evFree above, i.e. on the last kernel any OTHER lane's stream ran on this arena)
            if (h->arena) { H_TRY(hipStreamSynchronize(st)); (void)hipFree(h->arena); }
            h->arena = nullptr; h->arenaBytes = 0;
            for (;;) {
                const hipError_t e = hipMalloc(&h->arena, need);
                if (e == hipSuccess) break;
                (void)hipGetLastError();        // clear the sticky out-of-memory error
                h->arena = nullptr;
                if (e != hipErrorOutOfMemory || nS <= 256) { h->err = std::string("hybrid arena: hipMalloc(") + std::to_string(need) + "): " + hipGetErrorString(e); return RT_ERR_HIP; }
                nS = std::max<size_t>(nS / 2 / 256 * 256, 256);
                h->okSlots = nS;                // remembered: the next chunk and the next frame start here instead of failing the same allocations again
                T = nS * (size_t)SPP;
                capQ = std::min(std::max<size_t>((size_t)((double)T * h->ratioQ) + 4096, 4096), T * (size_t)qmax);
                capL = std::min<size_t>(std::min(std::max<size_t>((size_t)((double)T * h->ratioL) + 4096, 4096), T * (size_t)qmax * 4), 0xfffffff0u);
                need = bytes_for(nS, capQ, capL);
            }
            h->arenaBytes = need;
        }
        HybridBuf hb;
        {
            char *q = (char *)h->arena;
            auto take = [&](size_t bytes) { char *r = q; q += align_up(bytes, 256); return r; };
            hb.qO = (float4 *)take(capQ * 16); hb.qD = (float4 *)take(capQ * 16); hb.qDst = (uint32_t *)take(capQ * 4);
            hb.logT = (float *)take(capL * 4); hb.logTri = (int *)take(capL * 4); hb.logLim = (float *)take(capL * 4);
            hb.state = (uint32_t *)take(T * 4); hb.logBase = (uint32_t *)take(T * 4); hb.todo = (uint32_t *)take(T * 4); hb.recd = (uint32_t *)take(T * 4);
            hb.rad = (float4 *)take(T * 16);
            hb.sMotion = (float2 *)take(nS * 8); hb.sPos = (float4 *)take(nS * 16); hb.sNrm = (float4 *)take(nS * 16);
            hb.stgO = (float4 *)h->staging; hb.stgD = (float4 *)((char *)h->staging + stgEach);
        }
        hb.cnt = h->cnt; hb.qmax = qmax; hb.SPP = SPP; hb.check = h->check ? 1 : 0;
        hb.capQ = (uint32_t)std::min<size_t>(capQ, 0xfffffff0u); hb.capL = (uint32_t)capL;
        hb.slot0 = (uint32_t)slot0; hb.nS = (uint32_t)nS; hb.T = (uint32_t)T;

        H_TRY(hipMemsetAsync(hb.state, 0, T * 4, st));
        H_TRY(hipMemsetAsync(h->cnt, 0, C_WORDS * kCnt * sizeof(uint32_t), st));
        const unsigned gridV = (unsigned)(h->cus * 4);
        // Passes are queued in blocks without looking at their counts: all list lengths stay on the device, the kernels of a pass that has nothing
        // left return at once.  The host reads the counters once per block: first block = the passes a frame of this depth normally takes.
        int launched = 0, block = 3 + std::max(host.giBounces, 1);
        bool redo = false, finished = false;
        while (!finished) {
            for (int b = 0; b < block; ++b, ++launched) {
                H_TRY(hipMemsetAsync(h->cnt + C_OPEN * kCnt, 0, sizeof(uint32_t), st));
                H_TRY(hipMemsetAsync(h->cnt + C_REC * kCnt, 0, sizeof(uint32_t), st));
                H_TRY(hipMemsetAsync(h->cnt + C_TILE * kCnt, 0, sizeof(uint32_t), st));
                rt_stage_begin(ctx, ST_COMBINE, st);
                auto kfn = h->waves == 3 ? k_hybrid_shade<3> : h->waves == 5 ? k_hybrid_shade<5> : h->waves == 6 ? k_hybrid_shade<6> : k_hybrid_shade<4>;
                hipLaunchKernelGGL(kfn, dim3((unsigned)h->gridShade), dim3(256), 0, st, dFrame, hb, launched == 0 ? 0 : 1);
                hipLaunchKernelGGL(k_hybrid_note, dim3(1), dim3(1), 0, st, h->cnt);
                rt_stage_end(ctx, ST_COMBINE, 2, st);
                H_TRY(hipMemsetAsync(h->heads, 0, rt_wave_head_words() * sizeof(uint32_t), st));
                rt_stage_begin(ctx, ST_TRACE_GI, st);
                rt_wave_trace_closest_compact(st, h->cus, treeDepth, dFrame, host.sc, hb.qO, hb.qD, hb.qDst, &h->cnt[C_REC * kCnt], &h->cnt[C_FLAGS * kCnt], hb.capQ, hb.logT, hb.logTri, h->heads, h->check ? hb.capL : 0u);
                rt_stage_end(ctx, ST_TRACE_GI, 1, st);
                h->launches++;
                // the threads that recorded check their speculation; who failed is packed into the next pass's list
                H_TRY(hipMemsetAsync(h->cnt + C_TODO * kCnt, 0, sizeof(uint32_t), st));
                rt_stage_begin(ctx, ST_COMBINE, st);
                hipLaunchKernelGGL(k_hybrid_verify, dim3(gridV), dim3(256), 0, st, hb);
                rt_stage_end(ctx, ST_COMBINE, 1, st);
                if (h->debug) {
                    H_TRY(hipMemcpyAsync(h->hostCnt, h->cnt, C_WORDS * kCnt * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
                    H_TRY(hipStreamSynchronize(st));
                    fprintf(stderr, "[hybrid] slots %zu..%zu pass %d: %u threads recorded %u queries, %u to shade again, log %u / %u, flags %u\n", slot0, slot0 + nS, launched,
                            h->hostCnt[C_OPEN * kCnt], h->hostCnt[C_REC * kCnt], h->hostCnt[C_TODO * kCnt], h->hostCnt[C_LOG * kCnt], hb.capL, h->hostCnt[C_FLAGS * kCnt]);
                }
            }
            H_TRY(hipMemcpyAsync(h->hostCnt, h->cnt, C_WORDS * kCnt * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
            H_TRY(hipStreamSynchronize(st));
            const uint32_t flags = h->hostCnt[C_FLAGS * kCnt];
            if (flags & 1u) { h->err = "hybrid: a sample needs more than " + std::to_string(qmax) + " mesh queries"; return RT_ERR_UNSUPPORTED; }
            if (flags & 0x78u) { h->err = "hybrid check (RT_HYBRID_CHECK): a store left its array, flags " + std::to_string(flags) + " (8 thread list, 16 queue, 32 log, 64 staging)"; return RT_ERR_STATE; }
            if (flags & 6u) {
                // a pass outgrew the queue or the log arena: it wrote nothing, what follows it is void.  Enlarge from what was asked for and render the chunk again.
                if (flags & 2u) h->ratioQ = std::max(h->ratioQ * 1.5, (double)h->hostCnt[C_MAXREC * kCnt] / (double)T * 1.25);
                if (flags & 4u) h->ratioL = std::max(h->ratioL * 1.5, (double)h->hostCnt[C_LOG * kCnt] / (double)T * 1.25);
                h->ratioQ = std::min(h->ratioQ, (double)qmax); h->ratioL = std::min(h->ratioL, 4.0 * (double)qmax);
                h->redone++;
                if (++attempts > 64) { h->err = "hybrid: the arena estimates do not settle"; return RT_ERR_STATE; }
                redo = true;
                break;
            }
            if (h->hostCnt[C_OPEN * kCnt] == 0u) { h->passes += h->hostCnt[C_PASSES * kCnt]; finished = true; break; }   // the last shading pass left nothing open
            if (launched > 4 * (int)qmax) { h->err = "hybrid: passes do not converge"; return RT_ERR_STATE; }
            block = 2;
        }
        if (redo) continue;     // same slot0, larger capacities
        if (!waited && evPrevDone) { H_TRY(hipStreamWaitEvent(st, evPrevDone, 0)); waited = true; }   // the resolve reads the previous frame's COLOR0
        rt_stage_begin(ctx, ST_RESOLVE, st);
        hipLaunchKernelGGL(k_hybrid_resolve, dim3((unsigned)((nS + 255) / 256)), dim3(256), 0, st, dFrame, tg, hb);
        rt_stage_end(ctx, ST_RESOLVE, 1, st);
        slot0 += nS;
    }
    H_TRY(hipGetLastError());
    return RT_OK;
}

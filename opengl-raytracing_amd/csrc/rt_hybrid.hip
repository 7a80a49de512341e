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
#include "rt_device_analytic.hpp"
#include "rt_wave.hpp"

#pragma clang fp contract(off)

using namespace rtd;

namespace {

constexpr uint32_t kDone = 0xffffffffu;
constexpr uint32_t kCnt = 32;   // words between two counters: each on its own 128-byte line (they are hit by one atomic per workgroup)
enum { ST_TRACE_GI = 6, ST_RESOLVE = 8, ST_COMBINE = 9 };   // stage ids shared with rt_wave.hip (rt_stage_name)

struct HybridBuf {
    float4 *o, *d;         // [qmax x T] ray queue
    uint32_t *idx;         //            dense list of the queue addresses recorded in this pass (cnt[1] entries)
    float *logT;           //            answers: t (inf on a miss) ...
    int *logTri;           //            ... and triangle (-1 on a miss)
    uint32_t *state;       // [T] answered queries | recorded-up-to << 16, or kDone
    float4 *rad;           // [T] final radiance of (pixel, sample)
    float2 *sMotion;       // [nS] from the sample-0 thread: rt.frag:94-101, 172-175
    float4 *sPos, *sNrm;   // [nS]
    uint32_t *todo, *recd; // [T] dense lists of threads: to shade in this pass (behind the first pass) / that recorded queries in this pass
    uint32_t *cnt;         // at [k * kCnt]: k = 0 threads that recorded (= entries of recd), 1 queries recorded (= entries of idx), 2 overflow, 3 entries of todo
    uint32_t slot0, nS, T, qmax;
    int SPP;
};

// Five waves per SIMD: 96 VGPRs and 704 bytes of scratch per lane.  The shading code is long and cold in most of its registers; measured on MI355X
// (1080p, 16 spp, 4 bounces, all shading passes of a frame): 53.8 / 41.1 / 38.0 / 36.4 / 39.1 ms at 2 (no bound: 225 VGPRs) / 3 / 4 / 5 / 6 waves per SIMD.
__global__ __launch_bounds__(256, 5) void k_hybrid_shade(const DevFrame *__restrict__ fr, HybridBuf hb, int listed) {
    // first pass of a chunk: thread = (sample, slot) in order; later passes: the threads k_hybrid_verify left to do, packed (whole waves of work
    // instead of the few lanes per wave whose speculation failed: the second pass ran with 0.61 of its lanes, the third with 0.03)
    const uint32_t gid = blockIdx.x * 256u + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t n = listed ? hb.cnt[3 * kCnt] : hb.T;
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
                {
                Replay R;
                R.known = kn;
                R.thread = tid; R.stride = hb.T; R.qmax = hb.qmax;
                R.o = hb.o; R.d = hb.d; R.logT = hb.logT; R.logTri = hb.logTri;
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
                if (R.overflow) atomicOr(&hb.cnt[2 * kCnt], 1u);
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
    }
    // Each wave's share of the two dense lists (threads that recorded; the queue addresses they recorded, query-major inside the wave's block) is
    // reserved with ONE atomic pair per WORKGROUP: a single counter word takes about 90 M atomics per second on MI355X, and one pair per wave
    // (224 k waves in the first pass of a 1080p / 16 spp chunk) was most of that pass's time.
    __shared__ uint32_t sOpen[4], sTot[4], sBase[2];
    const uint32_t wv = threadIdx.x >> 6;
    const unsigned long long om = __ballot(open);
    uint32_t total = recorded, most = recorded;
    for (int off = 32; off > 0; off >>= 1) {
        total += (uint32_t)__shfl_down((int)total, off, 64);
        most = max(most, (uint32_t)__shfl_down((int)most, off, 64));
    }
    if (lane == 0u) { sOpen[wv] = (uint32_t)__popcll(om); sTot[wv] = total; }
    __syncthreads();
    if (threadIdx.x == 0u) {
        const uint32_t nOpen = sOpen[0] + sOpen[1] + sOpen[2] + sOpen[3], nTot = sTot[0] + sTot[1] + sTot[2] + sTot[3];
        sBase[0] = nOpen ? atomicAdd(&hb.cnt[0], nOpen) : 0u;
        sBase[1] = nTot ? atomicAdd(&hb.cnt[1 * kCnt], nTot) : 0u;
    }
    __syncthreads();
    if (om == 0ull) return;
    uint32_t tbase = sBase[0], base = sBase[1];
    for (uint32_t k = 0; k < wv; ++k) { tbase += sOpen[k]; base += sTot[k]; }
    most = (uint32_t)__shfl((int)most, 0, 64);
    const unsigned long long lt = (1ull << lane) - 1ull;
    if (open) hb.recd[tbase + (uint32_t)__popcll(om & lt)] = tid;      // whose speculation the next pass checks
    for (uint32_t k = 0; k < most; ++k) {
        const unsigned long long m = __ballot(recorded > k);
        if (recorded > k) hb.idx[base + (uint32_t)__popcll(m & lt)] = (known + k) * hb.T + tid;
        base += (uint32_t)__popcll(m);
    }
}

// Between a traversal launch and the next shading pass: every thread that recorded queries checks its speculation against the traced answers.
// All of them misses (or hits behind the analytic scene's own): the thread is finished, its stored radiance final.  Otherwise the answers up to the
// first failed query that later rays were built from join the log, and the thread goes onto the dense list of the next pass.
__global__ __launch_bounds__(256) void k_hybrid_verify(HybridBuf hb) {
    const uint32_t j = blockIdx.x * 256u + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u;
    bool again = false;
    uint32_t tid = 0;
    if (j < hb.cnt[0]) {
        tid = hb.recd[j];
        const uint32_t st = hb.state[tid];
        const uint32_t kn = st & 0xffffu, rec = st >> 16;
        bool anyHit = false;
        uint32_t valid = rec;                      // answers [kn, valid) are answers to queries the true frame asks
        for (uint32_t q = kn; q < rec; ++q) {
            const size_t a = (size_t)q * hb.T + tid;
            if (hb.logTri[a] >= 0 && hb.logT[a] < hb.o[a].w) {        // a mesh hit in front of the analytic scene's: the speculation failed here
                anyHit = true;
                if (hb.d[a].w != 0.0f) { valid = q + 1u; break; }      // later rays were built from the wrong hit: what follows is void
            }
        }
        if (!anyHit) hb.state[tid] = kDone;        // every speculated miss was one
        else { hb.state[tid] = valid; again = true; }
    }
    __shared__ uint32_t sAgain[4], sBase;
    const uint32_t wv = threadIdx.x >> 6;
    const unsigned long long am = __ballot(again);
    if (lane == 0u) sAgain[wv] = (uint32_t)__popcll(am);
    __syncthreads();
    if (threadIdx.x == 0u) {
        const uint32_t nAgain = sAgain[0] + sAgain[1] + sAgain[2] + sAgain[3];
        sBase = nAgain ? atomicAdd(&hb.cnt[3 * kCnt], nAgain) : 0u;      // one atomic per workgroup (see k_hybrid_shade)
    }
    __syncthreads();
    if (!again) return;
    uint32_t base = sBase;
    for (uint32_t k = 0; k < wv; ++k) base += sAgain[k];
    hb.todo[base + (uint32_t)__popcll(am & ((1ull << lane) - 1ull))] = tid;
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
    // ONE arena per context, shared by the frame lanes (evFree) and allocated for what the frame needs: 96 GB of the 288 hold a 1080p / 16 spp / 4-bounce
    // frame (80 GB: 33 M threads x 54 queries x 44 B) in ONE chunk -- 30.5 / 29.4 / 28.5 ms per frame with 32 / 48 / 96 GB (three / two / one chunk: each chunk has
    // its own tail of nearly empty passes)
    size_t budgetBytes = (size_t)96 << 30;
    hipEvent_t evFree = nullptr;          // recorded after a frame's last kernel: the next frame (another lane's stream) waits for it before it touches the arena
    void *arena = nullptr;
    size_t arenaBytes = 0;
    uint32_t *cnt = nullptr, *heads = nullptr;
    uint32_t *hostCnt = nullptr;       // pinned
    unsigned long long passes = 0, launches = 0;
    bool debug = false;                // RT_HYBRID_DEBUG=1: per-pass counts on stderr
};

RtHybrid *rt_hybrid_create(int cus) {
    RtHybrid *h = new RtHybrid();
    h->cus = cus > 0 ? cus : 256;
    if (const char *e = getenv("RT_QUEUE_BUDGET_MB")) h->budgetBytes = (size_t)atoll(e) << 20;
    (void)hipEventCreateWithFlags(&h->evFree, hipEventDisableTiming);
    if (const char *e = getenv("RT_HYBRID_DEBUG")) h->debug = atoi(e) != 0;
    return h;
}
void rt_hybrid_destroy(RtHybrid *h) {
    if (!h) return;
    if (h->arena) (void)hipFree(h->arena);
    if (h->cnt) (void)hipFree(h->cnt);
    if (h->heads) (void)hipFree(h->heads);
    if (h->hostCnt) (void)hipHostFree(h->hostCnt);
    if (h->evFree) (void)hipEventDestroy(h->evFree);
    delete h;
}
const char *rt_hybrid_error(const RtHybrid *h) { return h->err.c_str(); }

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
    const size_t perThread = (size_t)qmax * (16 + 16 + 4 + 4 + 4) + 3 * 4 + 16, perSlot = (size_t)SPP * perThread + 8 + 16 + 16;
    size_t nS = std::min(nSlots, std::max<size_t>(h->budgetBytes / perSlot, 256));
    nS = std::max<size_t>(nS / 256 * 256, 256);
    const size_t Tmax = nS * (size_t)SPP;
    if (qmax > 0xfff0u) { h->err = "hybrid: query log too long"; return RT_ERR_UNSUPPORTED; }
    if (Tmax * qmax >= ((size_t)1 << 31)) {   // queue addresses are 31-bit in the traversal kernel
        nS = std::max<size_t>((((size_t)1 << 31) - 1) / ((size_t)qmax * SPP) / 256 * 256, 256);
    }
    const size_t T0 = nS * (size_t)SPP, Q = (size_t)qmax * T0;
    const size_t need = align_up(Q * 16, 256) * 2 + align_up(Q * 4, 256) * 3 + align_up(T0 * 4, 256) * 3 + align_up(T0 * 16, 256) + align_up(nS * 8, 256) +
                        align_up(nS * 16, 256) * 2 + 4096;
    if (h->arenaBytes < need) {
        if (h->arena) (void)hipFree(h->arena);
        h->arena = nullptr; h->arenaBytes = 0;
        H_TRY(hipMalloc(&h->arena, need));
        h->arenaBytes = need;
    }
    if (!h->cnt) {
        H_TRY(hipMalloc((void **)&h->cnt, 4 * kCnt * sizeof(uint32_t)));
        H_TRY(hipMalloc((void **)&h->heads, rt_wave_head_words() * sizeof(uint32_t)));
        H_TRY(hipHostMalloc((void **)&h->hostCnt, 4 * kCnt * sizeof(uint32_t)));
    }
    HybridBuf hb;
    {
        char *q = (char *)h->arena;
        auto take = [&](size_t bytes) { char *r = q; q += align_up(bytes, 256); return r; };
        hb.o = (float4 *)take(Q * 16); hb.d = (float4 *)take(Q * 16);
        hb.idx = (uint32_t *)take(Q * 4); hb.logT = (float *)take(Q * 4); hb.logTri = (int *)take(Q * 4);
        hb.state = (uint32_t *)take(T0 * 4); hb.todo = (uint32_t *)take(T0 * 4); hb.recd = (uint32_t *)take(T0 * 4); hb.rad = (float4 *)take(T0 * 16);
        hb.sMotion = (float2 *)take(nS * 8); hb.sPos = (float4 *)take(nS * 16); hb.sNrm = (float4 *)take(nS * 16);
    }
    hb.cnt = h->cnt; hb.qmax = qmax; hb.SPP = SPP;
    if (h->evFree) H_TRY(hipStreamWaitEvent(st, h->evFree, 0));   // the previous frame's resolve kernels (on another lane's stream) still read the arena
    bool waited = false;
    for (size_t slot0 = 0; slot0 < nSlots; slot0 += nS) {
        const size_t nSc = std::min(nS, nSlots - slot0), T = nSc * (size_t)SPP;
        hb.slot0 = (uint32_t)slot0; hb.nS = (uint32_t)nSc; hb.T = (uint32_t)T;
        // the queue of a smaller last chunk uses stride T as well: entries stay inside the arena (T <= T0)
        H_TRY(hipMemsetAsync(hb.state, 0, T * 4, st));
        uint32_t todoBound = 0;                        // upper bound of the dense list's length: the threads that recorded in the pass before
        for (int pass = 0;; ++pass) {
            if (pass > 4 * (int)qmax) { h->err = "hybrid: passes do not converge"; return RT_ERR_STATE; }
            // cnt[0..2] belong to the shading pass about to run; cnt[3] (entries of todo) was written by the verification just before it
            H_TRY(hipMemsetAsync(h->cnt, 0, 3 * kCnt * sizeof(uint32_t), st));
            rt_stage_begin(ctx, ST_COMBINE, st);
            const size_t threads = pass == 0 ? T : (size_t)todoBound;
            hipLaunchKernelGGL(k_hybrid_shade, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, dFrame, hb, pass == 0 ? 0 : 1);
            rt_stage_end(ctx, ST_COMBINE, 1, st);
            H_TRY(hipMemcpyAsync(h->hostCnt, h->cnt, 4 * kCnt * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
            H_TRY(hipStreamSynchronize(st));
            h->passes++;
            const uint32_t open = h->hostCnt[0], recorded = h->hostCnt[1 * kCnt], overflow = h->hostCnt[2 * kCnt];
            if (h->debug) fprintf(stderr, "[hybrid] slots %zu..%zu pass %d: %zu threads shaded, %u of them recorded %u queries\n", slot0, slot0 + nSc, pass,
                                  pass == 0 ? T : (size_t)h->hostCnt[3 * kCnt], open, recorded);
            if (overflow) { h->err = "hybrid: a sample needs more than " + std::to_string(qmax) + " mesh queries"; return RT_ERR_UNSUPPORTED; }
            if (open == 0) break;
            if (recorded == 0) { h->err = "hybrid: open queries but nothing recorded"; return RT_ERR_STATE; }
            H_TRY(hipMemsetAsync(h->heads, 0, rt_wave_head_words() * sizeof(uint32_t), st));
            rt_stage_begin(ctx, ST_TRACE_GI, st);
            // cnt[1] stays what the shading pass left there until the next pass clears it: the launch reads the list length from it
            rt_wave_trace_closest_indexed(st, h->cus, treeDepth, dFrame, host.sc, hb.idx, &h->cnt[1 * kCnt], hb.o, hb.d, hb.logT, hb.logTri, h->heads);
            rt_stage_end(ctx, ST_TRACE_GI, 1, st);
            h->launches++;
            // the threads that recorded check their speculation; who failed is packed into the next pass's list
            H_TRY(hipMemsetAsync(h->cnt + 3 * kCnt, 0, sizeof(uint32_t), st));
            rt_stage_begin(ctx, ST_COMBINE, st);
            hipLaunchKernelGGL(k_hybrid_verify, dim3((unsigned)((open + 255u) / 256u)), dim3(256), 0, st, hb);
            rt_stage_end(ctx, ST_COMBINE, 1, st);
            todoBound = open;
        }
        if (!waited && evPrevDone) { H_TRY(hipStreamWaitEvent(st, evPrevDone, 0)); waited = true; }   // the resolve reads the previous frame's COLOR0
        rt_stage_begin(ctx, ST_RESOLVE, st);
        hipLaunchKernelGGL(k_hybrid_resolve, dim3((unsigned)((nSc + 255) / 256)), dim3(256), 0, st, dFrame, tg, hb);
        rt_stage_end(ctx, ST_RESOLVE, 1, st);
    }
    H_TRY(hipGetLastError());
    if (h->evFree) H_TRY(hipEventRecord(h->evFree, st));
    return RT_OK;
}

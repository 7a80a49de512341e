// rt_mega.hip -- megakernel: one thread = one fragment invocation of shaders/rt/rt.frag:50-197.
//
// Used for the analytic scene (pure ALU, nothing to queue) and as the reference-shaped baseline
// for BVH scenes that the wavefront pipeline (rt_wave.hip) is measured and bit-compared against.
// A workgroup is one 16x16 tile; each wave is an 8x8 pixel block; the traversal stack lives in LDS.
#include "rt_device_analytic.hpp"
#include "rt_frame.hpp"

#pragma clang fp contract(off)

using namespace rtd;

namespace {

// Rays are traced where the shading code asks for them.
template <bool COUNT>
struct InlineTracer {
    const DevScene *sc;
    float eps, inf;
    StackEntry *stk;
    Work *w;
    // the megakernel is the reference-shaped baseline: it traces every ray the reference casts, needed or not
    __device__ __noinline__ bool shadow(int, int, V3 ro, V3 rd, float tMax, bool) {
        uint32_t f0 = w->fetches();
        bool r = bvh_anyhit<COUNT>(*sc, ro, rd, eps, tMax, stk, *w);
        if (COUNT) w->fetchShadow += w->fetches() - f0;
        return r;
    }
    __device__ __noinline__ bool closest(V3 ro, V3 rd, float &t, int &tri) { return bvh_closest<COUNT>(*sc, ro, rd, eps, inf, stk, t, tri, *w); }
    RT_DEV int gi(V3 ro, V3 rd, V3 &hp, V3 &hn) {
        float t;
        int tri;
        if (!closest(ro, rd, t, tri)) return 0;
        hp = ro + rd * t;
        hn = tri_normal(*sc, tri);
        return 1;
    }
    RT_DEV bool ao(int, V3 org, V3 dir, float radius) {
        float t;
        int tri;
        return closest(org, dir, t, tri) && t < radius;
    }
};

// Four waves per SIMD (128 VGPRs + scratch) where the LDS stack allows it: the fragment program is long and most of its registers are cold.  Measured
// on MI355X, 1080p: BVH scene at 4 spp 22.2 -> 15.0 ms per frame (15.6 at five waves), hybrid extension 16.2 -> 13.4, analytic scene 0.53 -> 0.55.
template <bool COUNT, int STACK>
__global__ __launch_bounds__(256, 4) void k_mega(const DevFrame *__restrict__ fr, Targets tg, unsigned long long *counters) {
    __shared__ StackEntry lds_stack[4 * STACK * 64];
    const RtUniforms &u = fr->u;
    const FrameGeom &g = fr->g;
    const int tid = threadIdx.x;
    int px, py;
    const bool live = pixel_of_slot(g, blockIdx.x, tid, px, py);
    const int slot = blockIdx.x * 256 + tid;
    Work w;
    work_zero(w);
    if (live) {
        Frag F;
        F.u = &u; F.sc = &fr->sc; F.fcx = (float)px + 0.5f; F.fcy = (float)py + 0.5f;
        F.stk = &lds_stack[(tid >> 6) * STACK * 64 + (tid & 63)];
        F.giBounces = fr->giBounces;
        F.frameIndex = u.frameIndex;
        const int SPP = max(u.spp, 1);
        const V3 camPos = ld3(u.camPos);
        const V3 dir = primaryDir(u, F.fcx, F.fcy);
        V3 frameSum = mk3(0.0f);
        V2 motionOut = mk2(0.0f, 0.0f);
        V4 gpos = mk4(0.0f, 0.0f, 0.0f, 0.0f), gnrm = mk4(0.0f, 0.0f, 0.0f, 0.0f);
        const V3 V = -dir;

        if (u.useBVH == 1) {
            InlineTracer<COUNT> tr;
            tr.sc = &fr->sc; tr.eps = u.eps; tr.inf = u.inf; tr.w = &w;
            tr.stk = &lds_stack[(tid >> 6) * STACK * 64 + (tid & 63)];
            const bool bvhOn = (u.nodeCount > 0 && u.triCount > 0);
            // The SPP primary rays of rt.frag:79-86 are identical (only the seed changes): trace once,
            // account SPP times.  Same for computeAO (frame = uFrameIndex for every s, rt.frag:116).
            Work w0 = w;
            float tHit = u.inf;
            int triHit = -1;
            bool hitAny = bvhOn && tr.closest(camPos, dir, tHit, triHit);
            if (!bvhOn && COUNT) w.raysClosest++;
            if (COUNT) {
                w.raysClosest += (w.raysClosest - w0.raysClosest) * (uint32_t)(SPP - 1);
                w.nodeFetch += (w.nodeFetch - w0.nodeFetch) * (uint32_t)(SPP - 1);
                w.triFetch += (w.triFetch - w0.triFetch) * (uint32_t)(SPP - 1);
                w.fetchPrimary += w.fetches() - w0.fetches();
            }
            if (hitAny) {
                const V3 hp = camPos + dir * tHit;
                const V3 hn = tri_normal(fr->sc, triHit);
                if (COUNT) w.hitPixels++;
                V2 prevNDC = ndcFromWorld(hp, u.prevViewProj), currNDC = ndcFromWorld(hp, u.currViewProj);
                motionOut = mk2(currNDC.x - prevNDC.x, currNDC.y - prevNDC.y);
                gpos = mk4(hp.x, hp.y, hp.z, 1.0f);
                V3 nn = normalize(hn);
                gnrm = mk4(nn.x, nn.y, nn.z, 0.0f);
                float ao = 1.0f;
                if (u.enableAO == 1) {
                    Work w1 = w;
                    ao = computeAO_BVH(tr, F, hp, hn, u.frameIndex);
                    if (COUNT) {
                        w.raysClosest += (w.raysClosest - w1.raysClosest) * (uint32_t)(SPP - 1);
                        w.nodeFetch += (w.nodeFetch - w1.nodeFetch) * (uint32_t)(SPP - 1);
                        w.triFetch += (w.triFetch - w1.triFetch) * (uint32_t)(SPP - 1);
                        w.fetchAO += w.fetches() - w1.fetches();
                    }
                }
                for (int s = 0; s < SPP; ++s) {
                    int seed = (int)((uint32_t)u.frameIndex * (uint32_t)SPP + (uint32_t)s);
                    frameSum = frameSum + shadeSampleBVH<InlineTracer<COUNT>, COUNT>(tr, F, hp, hn, V, seed, ao, w);
                }
            } else {
                V3 r = sky<COUNT>(F, dir, w);
                if (COUNT) w.envLookup += (u.useEnvMap == 1) ? (uint32_t)(SPP - 1) : 0u;
                for (int s = 0; s < SPP; ++s) frameSum = frameSum + r;
                if (u.cameraMoved == 1) motionOut = mk2(4.0f, 4.0f);
            }
        } else {
            // analytic scene: rt.frag:118-161, every sample traced (the primary is five primitives)
            for (int s = 0; s < SPP; ++s) {
                int seed = (int)((uint32_t)u.frameIndex * (uint32_t)SPP + (uint32_t)s);
                Hit h;
                bool hitAny = traceScene<COUNT>(F, camPos, dir, true, true, h, w, true);   // the analytic scene (+ the mesh in the hybrid extension)
                V3 radiance;
                if (hitAny) {
                    if (s == 0) {
                        if (COUNT) w.hitPixels++;
                        V2 prevNDC = ndcFromWorld(h.p, u.prevViewProj), currNDC = ndcFromWorld(h.p, u.currViewProj);
                        motionOut = mk2(currNDC.x - prevNDC.x, currNDC.y - prevNDC.y);
                        gpos = mk4(h.p.x, h.p.y, h.p.z, 1.0f);
                        V3 nn = normalize(h.n);
                        gnrm = mk4(nn.x, nn.y, nn.z, 0.0f);
                    }
                    radiance = shadeSampleAnalytic<COUNT>(F, h, V, seed, w);
                } else {
                    radiance = sky<COUNT>(F, dir, w);
                    if (u.cameraMoved == 1 && s == 0) motionOut = mk2(4.0f, 4.0f);
                }
                frameSum = frameSum + radiance;
            }
        }
        V3 curr = frameSum / (float)SPP;
        float uvx = ((float)px + 0.5f) / (float)g.W, uvy = ((float)py + 0.5f) / (float)g.H;   // rt_fullscreen.vert:44
        V2 taaMotion = (u.cameraMoved == 1) ? motionOut : mk2(0.0f, 0.0f);
        HistoryTex hist;
        hist.prev = tg.prev; hist.prevAll = tg.prevAll; hist.blockSlots = tg.blockSlots; hist.g = &g; hist.slot = slot;
        V4 taa = resolveTAA(u, curr, uvx, uvy, taaMotion, u.frameIndex, hist);
        tg.color[slot] = pack_half4(taa);
        tg.motion[slot] = pack_half2(motionOut);
        tg.gpos[slot] = pack_half4(gpos);
        tg.gnrm[slot] = pack_half4(gnrm);
    }
    if (COUNT) flush_work(w, counters);
}

template <bool COUNT>
hipError_t launch_depth(hipStream_t s, const DevFrame *frame, Targets tg, unsigned long long *counters, int stackDepth, int nLocalTiles) {
    dim3 grid((unsigned)nLocalTiles), block(256);
    if (stackDepth <= 16) hipLaunchKernelGGL((k_mega<COUNT, 16>), grid, block, 0, s, frame, tg, counters);
    else if (stackDepth <= 24) hipLaunchKernelGGL((k_mega<COUNT, 24>), grid, block, 0, s, frame, tg, counters);
    else hipLaunchKernelGGL((k_mega<COUNT, 32>), grid, block, 0, s, frame, tg, counters);
    return hipGetLastError();
}

}  // namespace

namespace rtl {
hipError_t launch_mega(hipStream_t s, const DevFrame *frame, Targets tg, unsigned long long *counters, bool count, int stackDepth,
                       int nLocalTiles) {
    if (nLocalTiles <= 0) return hipSuccess;
    return count ? launch_depth<true>(s, frame, tg, counters, stackDepth, nLocalTiles)
                 : launch_depth<false>(s, frame, tg, counters, stackDepth, nLocalTiles);
}
}  // namespace rtl

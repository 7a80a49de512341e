// rt_device_shade.hpp -- device-side restatement of the reference fragment program's building
// blocks (shaders/rt/rt_common.glsl, rt_materials.glsl, rt_scene_analytic.glsl, rt_bvh.glsl,
// rt_lighting.glsl, rt_taa.glsl), organised for two consumers:
//   * the megakernel (one thread = one fragment invocation, rays traced in place), and
//   * the wavefront pipeline, where the SAME shading code runs three times with different
//     "tracer" policies: emit rays into queues, emit second-generation rays, combine results.
// BVH traversal here is not the reference's loop: it walks a 64-byte two-child node (both child
// boxes in the parent), keeps (child, entry distance) pairs on an LDS stack and descends into the
// near child without a push/pop round trip.  It visits nodes and triangles in exactly the order
// of rt_bvh.glsl:193-304, so hits (and ties) are identical; see DESIGN.md "Traversal equivalence".
#pragma once
#include "../../include/rt_mi355.h"
#include "rt_device_math.hpp"

#pragma clang fp contract(off)

namespace rtd {

// ---------------------------------------------------------------------------------------------
// Scene in HBM.
//   wnodes: 4 x float4 per INNER node: [Lmin.xyz, leftRef] [Lmax.xyz, rightRef] [Rmin.xyz, -] [Rmax.xyz, -]
//   tris  : 3 x float4 per triangle  : [v0.xyz, -] [e1.xyz, -] [e2.xyz, -]      (bvh.cpp:187-204 order)
//   child reference: >= 0 -> inner node index;  < 0 -> leaf, v = -ref-1, first = v >> 3, count = (v & 7) + 1
//   env   : 6 faces of RGBA8 (GL face order), envSize^2 texels each
//   w4    : 8 x float4 per 4-wide node (two binary levels collapsed; any-hit rays only, where visiting order is
//           free), component-wise: [min.x of children 0..3] [min.y] [min.z] [max.x] [max.y] [max.z] [ref 0..3] [-]: 7 loads;
//           ref == RT_NO_CHILD (and a NaN box) for an absent child
#define RT_NO_CHILD 0x7fffffff
struct DevScene {
    const float4 *wnodes;
    const float4 *w4;
    const float4 *q4;        // RT_QNODES: w4 with quantised child boxes, 4 x 16 bytes per node (null: not built), see rt_upload_bvh
    const float4 *leafBox;   // RT_QNODES: 2 x float4 per leaf: its exact box [lo.xyz hi.x][hi.yz - -], at index (first pair record * leafBoxMagic) >> 32
    uint32_t leafBoxMagic;   // ceil(2^32 / R), R = fewest pair records of a leaf (0: index = first pair record), see rt_upload_bvh
    const float4 *wnodesW;   // wnodes with pair-record leaf references (wavefront closest-hit kernels)
    const float4 *wF;        // fused closest-hit records (round 5): 8 x float4 per even-level inner node = the wnodesW records of its two children, see rt_upload_bvh (null: not built)
    const float4 *iN2;       // implicit two-child records (round 5): 3 x float4 per inner node [Lmin.xyz Lmax.x][Lmax.yz Rmin.xy][Rmin.z Rmax.xyz] at d - popcount(p) + (p << (implD - d)); null unless every leaf sits at depth implD
    const float4 *iPairs;    // `pairs` in leaf order: leaf p owns the records from p * implR on, the spare word of the first holds the leaf's triangle count
    int implD, implR;
    const float4 *iN4;       // implicit four-wide any-hit records: 6 x float4 per even-depth inner node [min.x x4][min.y][min.z][max.x][max.y][max.z] at the node's pre-order position
    const float4 *iQ4;       // the same quantised: 3 x float4 [origin.xyz, exponents][lo.x lo.y lo.z hi.x][hi.y hi.z - -] (null: the exact form is walked)
    const float4 *iLeafBox;  // with iQ4: 2 x float4 per leaf, its exact box, at the leaf's ordinal
    const float4 *pairs;     // 5 x float4 per PAIR of triangles of a leaf: [v0 e1 e2][v0 e1 e2][index of the first][-], see rt_upload_bvh
    const float4 *tris;
    const uchar4 *env;
    int envSize;
    int envFilter;   // 0: bilinear weights in exact fp32 (default); 1: texel coordinates rounded to 1/256 of a texel first (RtExtension.envFilter)
    int rootRef;
    int rootRef4;
    int rootRefW;
    int hasBVH;
    float rootMin[3], rootMax[3];
    int anyStack;   // per-lane stack entries the any-hit walk of w4 can need (3 per level of the 4-wide tree)
};

struct Work {   // RtCounters, per lane
    uint32_t raysClosest, raysShadow, raysAnalytic, nodeFetch, triFetch, envLookup, hitPixels;
    uint32_t fetchPrimary, fetchShadow, fetchAO;
    RT_DEV uint32_t fetches() const { return nodeFetch + triFetch; }
};
RT_DEV void work_zero(Work &w) {
    w.raysClosest = w.raysShadow = w.raysAnalytic = w.nodeFetch = w.triFetch = w.envLookup = w.hitPixels = 0;
    w.fetchPrimary = w.fetchShadow = w.fetchAO = 0;
}

struct Hit { float t; V3 p; V3 n; int mat; };   // rt_common.glsl:39-44

// ---------------------------------------------------------------------------------------------
// rt_common.glsl
RT_DEV uint32_t hash2(uint32_t vx, uint32_t vy) {   // :57-63
    vx = vx * 1664525u + 1013904223u;
    vy = vy * 1664525u + 1013904223u;
    vx ^= vy >> 16;
    vy ^= vx << 5;
    vx = vx * 1664525u + 1013904223u;
    vy = vy * 1664525u + 1013904223u;
    return vx ^ vy;
}
RT_DEV uint32_t f2uint(float p) { return (uint32_t)fminr(fmaxr(p, 0.0f), 4294967040.0f); }   // uvec2(vec2)
RT_DEV uint32_t rand_bits(float px, float py, int frame) {
    uint32_t fx = (uint32_t)frame, fy = (uint32_t)frame * 1663u;
    return hash2(f2uint(px) ^ fx, f2uint(py) ^ fy);
}
RT_DEV float randr(float px, float py, int frame) { return (float)rand_bits(px, py, frame) / 4294967296.0f; }   // :75-77
RT_DEV float epsForDist(float d) { return fmaxr(1e-4f, 1e-3f * d); }   // :88-90
RT_DEV float halton(int i, int b) {   // :106-116
    float f = 1.0f, r = 0.0f;
    int n = i;
    while (n > 0) {
        f /= (float)b;
        r += f * (float)(n % b);
        n /= b;
    }
    return r;
}
RT_DEV V2 concentricSample(float pi, V2 u) {   // :144-159
    float a = 2.0f * u.x - 1.0f;
    float b = 2.0f * u.y - 1.0f;
    float r, phi;
    if (a == 0.0f && b == 0.0f) { r = 0.0f; phi = 0.0f; }
    else if (__builtin_fabsf(a) > __builtin_fabsf(b)) { r = a; phi = (pi / 4.0f) * (b / a); }
    else { r = b; phi = (pi / 2.0f) - (pi / 4.0f) * (a / b); }
    float s, c;
    sincosr(phi, s, c);
    return mk2(r * c, r * s);
}
RT_DEV V2 ndcFromWorld(V3 p, const float *VP) {   // :175-179
    float cx = __builtin_fmaf(VP[8], p.z, __builtin_fmaf(VP[4], p.y, VP[0] * p.x)) + VP[12];
    float cy = __builtin_fmaf(VP[9], p.z, __builtin_fmaf(VP[5], p.y, VP[1] * p.x)) + VP[13];
    float cw = __builtin_fmaf(VP[11], p.z, __builtin_fmaf(VP[7], p.y, VP[3] * p.x)) + VP[15];
    float w = fmaxr(cw, 1e-6f);
    return mk2(cx / w, cy / w);
}

// ---------------------------------------------------------------------------------------------
// rt_bvh.glsl: slab test :124-134, Moller-Trumbore :154-170.
RT_DEV bool slab(V3 ro, V3 rdInv, V3 bmin, V3 bmax, float &tminOut) {
    V3 t0 = (bmin - ro) * rdInv;
    V3 t1 = (bmax - ro) * rdInv;
    float tmin = fmaxr(fmaxr(fminr(t0.x, t1.x), fminr(t0.y, t1.y)), fmaxr(fminr(t0.z, t1.z), 0.0f));
    float tmax = fminr(fminr(fmaxr(t0.x, t1.x), fmaxr(t0.y, t1.y)), fmaxr(t0.z, t1.z));
    tminOut = tmin;
    return tmax >= tmin;
}
RT_DEV bool tri_hit(V3 ro, V3 rd, V3 v0, V3 e1, V3 e2, float eps, float tMax, float &t) {
    V3 pvec = cross(rd, e2);
    float det = dot(e1, pvec);
    if (__builtin_fabsf(det) < 1e-8f) return false;
    float invDet = 1.0f / det;
    V3 tvec = ro - v0;
    float u = dot(tvec, pvec) * invDet;
    if (u < 0.0f || u > 1.0f) return false;
    V3 qvec = cross(tvec, e1);
    float v = dot(rd, qvec) * invDet;
    if (v < 0.0f || u + v > 1.0f) return false;
    float tt = dot(e2, qvec) * invDet;
    if (tt < eps || tt > tMax) return false;
    t = tt;
    return true;
}
RT_DEV V3 f4xyz(float4 v) { return mk3(v.x, v.y, v.z); }

// Per-lane traversal stack in LDS: entry e of this lane lives at stk[e * 64] (uint2 = {ref, tmin bits}),
// so one wave-wide push/pop is a conflict-free ds_write_b64 / ds_read_b64.
typedef uint2 StackEntry;

// Closest hit.  Returns true and (tBest, triBest) when something was hit; identical visiting order to
// traceBVH (rt_bvh.glsl:193-243): near child first, far child deferred, "tminBox > best" cull on pop,
// ties (tt == best) overwrite.  Reference-unit counters: one nodeFetch per node popped (culled or
// not), two more per inner node expanded, one triFetch per triangle tested.
template <bool COUNT>
RT_DEV bool bvh_closest(const DevScene &sc, V3 ro, V3 rd, float eps, float inf, StackEntry *stk, float &tBest, int &triBest,
                        Work &w) {
    if (COUNT) w.raysClosest++;
    tBest = inf;
    triBest = -1;
    if (!sc.hasBVH) return false;
    V3 rdInv = mk3(1.0f / rd.x, 1.0f / rd.y, 1.0f / rd.z);
    float tminCur;
    if (COUNT) w.nodeFetch++;
    if (!slab(ro, rdInv, ld3(sc.rootMin), ld3(sc.rootMax), tminCur) || tminCur > tBest) return false;
    int ref = sc.rootRef;
    int sp = 0;
    for (;;) {
        if (ref >= 0) {
            const float4 *n = sc.wnodes + (size_t)ref * 4;
            float4 a = n[0], b = n[1], c = n[2], d = n[3];
            if (COUNT) w.nodeFetch += 2;
            float tL, tR;
            bool hitL = slab(ro, rdInv, f4xyz(a), f4xyz(b), tL) && tL <= tBest;
            bool hitR = slab(ro, rdInv, f4xyz(c), f4xyz(d), tR) && tR <= tBest;
            int refL = (int)f2u(a.w), refR = (int)f2u(b.w);
            if (hitL && hitR) {
                bool leftFirst = tL < tR;
                StackEntry e;
                e.x = (uint32_t)(leftFirst ? refR : refL);
                e.y = f2u(leftFirst ? tR : tL);
                stk[sp * 64] = e;
                sp++;
                ref = leftFirst ? refL : refR;
                if (COUNT) w.nodeFetch++;   // the reference pops the near child right back
                continue;
            }
            if (hitL || hitR) {
                ref = hitL ? refL : refR;
                if (COUNT) w.nodeFetch++;
                continue;
            }
        } else {
            int v = -ref - 1;
            int first = v >> 3, count = (v & 7) + 1;
            for (int i = 0; i < count; ++i) {
                const float4 *t = sc.tris + (size_t)(first + i) * 3;
                float4 p0 = t[0], p1 = t[1], p2 = t[2];
                if (COUNT) w.triFetch++;
                float tt;
                if (tri_hit(ro, rd, f4xyz(p0), f4xyz(p1), f4xyz(p2), eps, tBest, tt)) { tBest = tt; triBest = first + i; }
            }
        }
        // pop: skip entries whose box starts beyond the current best (rt_bvh.glsl:208)
        bool found = false;
        while (sp > 0) {
            sp--;
            StackEntry e = stk[sp * 64];
            if (COUNT) w.nodeFetch++;
            if (u2f(e.y) > tBest) continue;
            ref = (int)e.x;
            found = true;
            break;
        }
        if (!found) break;
    }
    return tBest < inf;
}

// Any hit within [eps, tMax]: traceBVHShadow, rt_bvh.glsl:260-304.
template <bool COUNT>
RT_DEV bool bvh_anyhit(const DevScene &sc, V3 ro, V3 rd, float eps, float tMax, StackEntry *stk, Work &w) {
    if (COUNT) w.raysShadow++;
    if (!sc.hasBVH) return false;
    V3 rdInv = mk3(1.0f / rd.x, 1.0f / rd.y, 1.0f / rd.z);
    float tminCur;
    if (COUNT) w.nodeFetch++;
    if (!slab(ro, rdInv, ld3(sc.rootMin), ld3(sc.rootMax), tminCur) || tminCur > tMax) return false;
    int ref = sc.rootRef;
    int sp = 0;
    for (;;) {
        if (ref >= 0) {
            const float4 *n = sc.wnodes + (size_t)ref * 4;
            float4 a = n[0], b = n[1], c = n[2], d = n[3];
            if (COUNT) w.nodeFetch += 2;
            float tL, tR;
            bool hitL = slab(ro, rdInv, f4xyz(a), f4xyz(b), tL) && tL <= tMax;
            bool hitR = slab(ro, rdInv, f4xyz(c), f4xyz(d), tR) && tR <= tMax;
            int refL = (int)f2u(a.w), refR = (int)f2u(b.w);
            if (hitL && hitR) {
                bool leftFirst = tL < tR;
                StackEntry e;
                e.x = (uint32_t)(leftFirst ? refR : refL);
                e.y = 0u;
                stk[sp * 64] = e;
                sp++;
                ref = leftFirst ? refL : refR;
                if (COUNT) w.nodeFetch++;
                continue;
            }
            if (hitL || hitR) {
                ref = hitL ? refL : refR;
                if (COUNT) w.nodeFetch++;
                continue;
            }
        } else {
            int v = -ref - 1;
            int first = v >> 3, count = (v & 7) + 1;
            for (int i = 0; i < count; ++i) {
                const float4 *t = sc.tris + (size_t)(first + i) * 3;
                float4 p0 = t[0], p1 = t[1], p2 = t[2];
                if (COUNT) w.triFetch++;
                float tt;
                if (tri_hit(ro, rd, f4xyz(p0), f4xyz(p1), f4xyz(p2), eps, tMax, tt)) return true;
            }
        }
        if (sp == 0) break;
        sp--;
        ref = (int)stk[sp * 64].x;
        if (COUNT) w.nodeFetch++;
    }
    return false;
}

// Geometric normal of triangle `tri` (rt_bvh.glsl:168) -- computed once for the final hit.
RT_DEV V3 tri_normal(const DevScene &sc, int tri) {
    const float4 *t = sc.tris + (size_t)tri * 3;
    return normalize(cross(f4xyz(t[1]), f4xyz(t[2])));
}

// ---------------------------------------------------------------------------------------------
// EXTENSION (hybrid scene, staged): replay state of one (pixel, sample) thread of rt_hybrid.hip.  The analytic shading code asks for mesh
// hits through ONE function (traceScene); staged, that function answers from a log of earlier passes and records what it cannot answer yet
// into a ray queue that a persistent traversal launch then traces.  Queries are numbered in program order (only those whose ray meets
// the mesh's root box count).  Round 4: the log is the thread's own contiguous block of a bump arena (entry q at logT[q] / logTri[q]), and an
// open query is written to the WORKGROUP's staging area (entry q of thread `slot` at [q * 256 + slot]), from which the end of the shading pass
// packs what was really recorded into dense arrays -- memory follows the recorded queries, not qmax x threads.
struct Replay {
    uint32_t q = 0;           // queries met so far in this pass
    uint32_t known = 0;       // queries answered by earlier passes
    uint32_t recEnd = 0;      // one past the last query recorded in this pass
    uint32_t pending = 0;     // queries without an answer met in this pass: recorded, and answered "no hit" on speculation
    bool overflow = false;    // more than qmax queries
    uint32_t slot = 0, qmax = 0;
    float4 *o = nullptr, *d = nullptr;   // staging area of this thread's workgroup
    const float *logT = nullptr;         // this thread's log
    const int *logTri = nullptr;
};

// ---------------------------------------------------------------------------------------------
// Per-fragment context.
struct Frag {
    const RtUniforms *u;   // kernel-argument copy (uniform across the grid)
    const DevScene *sc;
    float fcx, fcy;        // gl_FragCoord.xy
    // EXTENSION (hybrid scene, RT_SCENE_HYBRID): this lane's traversal stack, so that the analytic scene query can also walk the BVH;
    // bounces of the analytic / hybrid GI path (1 = the reference).  Unused in the reference's two modes.
    StackEntry *stk = nullptr;
    Replay *rp = nullptr;   // staged hybrid pipeline (rt_hybrid.hip); null in the megakernel
    int giBounces = 1;
    // uFrameIndex of THIS fragment's frame.  The reference's value for a single frame; with frame batching (rt_render_frames: K frames of
    // a static camera in one set of launches) the frames of a batch differ in it -- and in the jitter, see primaryDirJ -- and in nothing else.
    int frameIndex = 0;
};

// texture(uEnvMap, dir): face selection per the GL cube-map table, LINEAR, CLAMP_TO_EDGE, not seamless.
template <bool COUNT>
RT_DEV V3 texture_cube(const DevScene &sc, V3 d, Work &w) {
    if (COUNT) w.envLookup++;
    float ax = __builtin_fabsf(d.x), ay = __builtin_fabsf(d.y), az = __builtin_fabsf(d.z);
    int face;
    float scv, tcv, ma;
    if (ax >= ay && ax >= az) { ma = ax; if (d.x >= 0.0f) { face = 0; scv = -d.z; tcv = -d.y; } else { face = 1; scv = d.z; tcv = -d.y; } }
    else if (ay >= az)        { ma = ay; if (d.y >= 0.0f) { face = 2; scv = d.x; tcv = d.z; } else { face = 3; scv = d.x; tcv = -d.z; } }
    else                      { ma = az; if (d.z >= 0.0f) { face = 4; scv = d.x; tcv = -d.y; } else { face = 5; scv = -d.x; tcv = -d.y; } }
    float s = 0.5f * (scv / ma + 1.0f);
    float t = 0.5f * (tcv / ma + 1.0f);
    const int N = sc.envSize;
    float fu = s * (float)N - 0.5f, fv = t * (float)N - 0.5f;
    if (sc.envFilter == 1) {   // fixed-point texel coordinates, 8 fractional bits, round to nearest (see RtExtension.envFilter)
        fu = __builtin_floorf(fu * 256.0f + 0.5f) * 0.00390625f;
        fv = __builtin_floorf(fv * 256.0f + 0.5f) * 0.00390625f;
    }
    float flu = __builtin_floorf(fu), flv = __builtin_floorf(fv);
    float a = fu - flu, b = fv - flv;
    int i0 = (int)flu, j0 = (int)flv;
    int i1 = i0 + 1, j1 = j0 + 1;
    i0 = min(max(i0, 0), N - 1); i1 = min(max(i1, 0), N - 1);
    j0 = min(max(j0, 0), N - 1); j1 = min(max(j1, 0), N - 1);
    const uchar4 *F = sc.env + (size_t)face * N * N;
    uchar4 c00 = F[(size_t)j0 * N + i0], c10 = F[(size_t)j0 * N + i1], c01 = F[(size_t)j1 * N + i0], c11 = F[(size_t)j1 * N + i1];
    V3 t00 = mk3((float)c00.x / 255.0f, (float)c00.y / 255.0f, (float)c00.z / 255.0f);
    V3 t10 = mk3((float)c10.x / 255.0f, (float)c10.y / 255.0f, (float)c10.z / 255.0f);
    V3 t01 = mk3((float)c01.x / 255.0f, (float)c01.y / 255.0f, (float)c01.z / 255.0f);
    V3 t11 = mk3((float)c11.x / 255.0f, (float)c11.y / 255.0f, (float)c11.z / 255.0f);
    float w00 = (1.0f - a) * (1.0f - b), w10 = a * (1.0f - b), w01 = (1.0f - a) * b, w11 = a * b;
    return t00 * w00 + t10 * w10 + t01 * w01 + t11 * w11;
}
template <bool COUNT>
RT_DEV V3 sky(const Frag &F, V3 dir, Work &w) {   // rt_scene_analytic.glsl:211-223
    if (F.u->useEnvMap == 1) return texture_cube<COUNT>(*F.sc, dir, w) * F.u->envIntensity;
    float t = clampr(0.5f * (dir.y + 1.0f), 0.0f, 1.0f);
    return mix(mk3(0.6f, 0.7f, 0.9f) * 0.3f, mk3(0.1f, 0.15f, 0.3f) * 0.3f, 1.0f - t);
}

// ---------------------------------------------------------------------------------------------
// rt_lighting.glsl pieces that do not trace.
struct MaterialProps { V3 albedo; float specStrength; float gloss; int type; float ior; };   // rt_materials.glsl:36-42

RT_DEV V3 shadeLambertPhong(float pi, V3 N, V3 V, V3 L, V3 Li, V3 albedo, float specStrength, float gloss) {   // :78-98
    float ndl = fmaxr(dot(N, L), 0.0f);
    if (ndl <= 0.0f) return mk3(0.0f);
    V3 diffuse = albedo * (ndl / pi);
    V3 spec = mk3(0.0f);
    if (specStrength > 0.0f) {
        V3 H = normalize(L + V);
        float ndh = fmaxr(dot(N, H), 0.0f);
        float phong = powr(ndh, gloss);
        spec = (specStrength * phong) * mk3(1.0f);
    }
    return (diffuse + spec) * Li;
}
RT_DEV V3 skyDirect(const RtUniforms &u, V3 hn, const MaterialProps &mat) {   // :156-169
    if (u.skyEnabled == 0) return mk3(0.0f);
    V3 N = normalize(hn);
    V3 U = normalize(ld3(u.skyUpDir));
    float ndl = fmaxr(dot(N, U), 0.0f);
    if (ndl <= 0.0f) return mk3(0.0f);
    V3 Li = ld3(u.skyColor) * u.skyIntensity;
    return mat.albedo * (ndl / u.pi) * Li;
}
RT_DEV void buildONB(V3 N, V3 &T, V3 &B) {   // :227-231
    V3 up = (__builtin_fabsf(N.y) < 0.99f) ? mk3(0.0f, 1.0f, 0.0f) : mk3(1.0f, 0.0f, 0.0f);
    T = normalize(cross(up, N));
    B = cross(N, T);
}
RT_DEV V3 sampleHemisphereCosine(float pi, V3 N, V2 uu) {   // :251-266
    float phi = 2.0f * pi * uu.x;
    float r = __builtin_sqrtf(uu.y);
    float sn, cs;
    sincosr(phi, sn, cs);
    float x = r * cs;
    float z = r * sn;
    float y = __builtin_sqrtf(fmaxr(0.0f, 1.0f - uu.y));
    V3 T, B;
    buildONB(normalize(N), T, B);
    return normalize(x * T + z * B + y * N);
}
RT_DEV V2 cpOffset(float px, float py, int frame) {   // :280-289
    float hx = randr(px, py, (int)((uint32_t)frame * 911u));
    float hy = randr(py, px, (int)((uint32_t)frame * 577u));
    float lx = halton(frame + 1, 2), ly = halton(frame + 1, 3);   // ld2(frame), rt_common.glsl:127-129
    return mk2(fractr(hx + lx), fractr(hy + ly));
}
RT_DEV V3 kLightN() { return normalize(mk3(0.0f, -1.0f, 0.2f)); }   // :30
RT_DEV void lightFrame(V3 &t, V3 &b) {   // :355-357
    V3 n = kLightN();
    t = normalize(__builtin_fabsf(n.y) < 0.99f ? cross(n, mk3(0.0f, 1.0f, 0.0f)) : cross(n, mk3(1.0f, 0.0f, 0.0f)));
    b = cross(n, t);
}

// One disk-light sample i of directLight / directLightBVH (:363-385 / :422-443) up to the visibility
// ray: returns the light point xL and the shadow ray occludedToward(p, xL) would cast (:49-54).
struct DiskSample { V3 xL; V3 L; float geom; V3 ro; V3 rd; float tMax; };
RT_DEV DiskSample diskSample(const Frag &F, V3 hp, V3 N, int frame, int i, V2 rot, V3 lt, V3 lb) {
    const float kLightRadius = 1.2f;
    const V3 kLightCenter = mk3(0.0f, 5.0f, -3.0f);
    float fi = (float)i, fk = (float)(31 * i + 7);
    V2 uu = mk2(randr(F.fcx + fi, F.fcy + fi, frame), randr(F.fcy + fk, F.fcx + fk, frame));
    uu = mk2(fractr(uu.x + rot.x), fractr(uu.y + rot.y));
    V2 cd = concentricSample(F.u->pi, uu);
    V2 d = mk2(cd.x * kLightRadius, cd.y * kLightRadius);
    DiskSample s;
    s.xL = kLightCenter + lt * d.x + lb * d.y;
    s.L = normalize(s.xL - hp);
    float ndl = fmaxr(dot(N, s.L), 0.0f);
    float cosThetaL = fmaxr(dot(-kLightN(), s.L), 0.0f);
    float r2 = fmaxr(dot(s.xL - hp, s.xL - hp), 1e-4f);
    s.geom = (ndl * cosThetaL) / r2;
    // occludedToward(h.p, xL)
    s.rd = normalize(s.xL - hp);
    float maxT = length(s.xL - hp);
    float e = epsForDist(maxT);
    s.ro = hp + s.rd * e;
    s.tMax = maxT - e;
    return s;
}

// Segments of one BVH-mode sample; a tracer policy maps (segment, k) to a queue slot.
enum { SEG_DIRECT = 0, SEG_GI_DIRECT = 1 };

// directLightBVH (:405-460) over a tracer policy T:
//   bool T::shadow(int seg, int k, V3 ro, V3 rd, float tMax, bool matters)   k = 0..3 disk, 4 sun, 5 point
// `matters` is false when the visibility cannot reach the output: the reference casts the disk-sample
// ray even when geom == 0 (rt_lighting.glsl:437-438), but then Li = kLightCol*0*vis = 0 for either
// answer.  A tracer may skip such a ray and return anything.
template <class T>
RT_DEV V3 directLightBVH(T &tr, const Frag &F, int seg, V3 hp, V3 hn, int frame, V3 Vdir) {
    const RtUniforms &u = *F.u;
    V3 N = normalize(hn);
    V3 sum = mk3(0.0f);
    const V3 albedo = mk3(0.85f);
    const float specStrength = 0.25f, gloss = 32.0f;
    V3 lt, lb;
    lightFrame(lt, lb);
    V2 rot = cpOffset(F.fcx, F.fcy, F.frameIndex);
    V3 V = normalize(Vdir);
    for (int i = 0; i < 4; ++i) {   // SOFT_SHADOW_SAMPLES
        DiskSample s = diskSample(F, hp, N, frame, i, rot, lt, lb);
        float vis = tr.shadow(seg, i, s.ro, s.rd, s.tMax, s.geom != 0.0f) ? 0.0f : 1.0f;
        V3 Li = mk3(18.0f) * s.geom * vis;
        sum = sum + shadeLambertPhong(u.pi, N, V, s.L, Li, albedo, specStrength, gloss);
    }
    sum = sum / 4.0f;
    MaterialProps fakeMat;
    fakeMat.albedo = albedo; fakeMat.specStrength = specStrength; fakeMat.gloss = gloss; fakeMat.type = 0; fakeMat.ior = 1.0f;
    // sunDirect :114-144
    V3 sun = mk3(0.0f);
    if (u.sunEnabled != 0) {
        V3 Vs = normalize(V);   // sunDirect re-normalises the already normalised V (rt_lighting.glsl:119)
        V3 L = normalize(-ld3(u.sunDir));
        float ndl = fmaxr(dot(N, L), 0.0f);
        if (ndl > 0.0f) {
            float maxT = 1000.0f;
            float e = epsForDist(maxT);
            V3 origin = hp + N * e;
            bool blocked = tr.shadow(seg, 4, origin, L, maxT - e, true);
            if (!blocked) sun = shadeLambertPhong(u.pi, N, Vs, L, ld3(u.sunColor) * u.sunIntensity, albedo, specStrength, gloss);
        }
    }
    sum = sum + sun;
    sum = sum + skyDirect(u, hn, fakeMat);
    // pointDirect :181-214
    V3 pt = mk3(0.0f);
    if (u.pointLightEnabled != 0) {
        V3 Vp = normalize(V);   // pointDirect does the same (rt_lighting.glsl:186)
        V3 toL = ld3(u.pointLightPos) - hp;
        float dist2 = dot(toL, toL);
        if (dist2 > 1e-6f) {
            float dist = __builtin_sqrtf(dist2);
            V3 L = toL / dist;
            float ndl = fmaxr(dot(N, L), 0.0f);
            if (ndl > 0.0f) {
                float e = epsForDist(dist);
                V3 origin = hp + L * e;
                bool blocked = tr.shadow(seg, 5, origin, L, dist - e, true);
                if (!blocked) {
                    V3 Li = ld3(u.pointLightColor) * (u.pointLightIntensity / fmaxr(dist2, 1e-4f));
                    pt = shadeLambertPhong(u.pi, N, Vp, L, Li, albedo, specStrength, gloss);
                }
            }
        }
    }
    sum = sum + pt;
    return sum;
}

// oneBounceGIBVH (:515-561):  int T::gi(V3 ro, V3 rd, V3 &hp, V3 &hn)  -> 1 hit, 0 miss, -1 "not known yet"
template <class T, bool COUNT>
RT_DEV V3 oneBounceGIBVH(T &tr, const Frag &F, V3 hp0, V3 hn0, int frame, int seed, Work &w) {
    const RtUniforms &u = *F.u;
    const V3 albedo0 = mk3(0.85f);
    const float MAX_GI_LUM = 8.0f, MIN_COS_THETA = 0.1f;
    float o19 = (float)(int)((uint32_t)seed * 19u), o41 = (float)(int)((uint32_t)seed * 41u);
    V2 uu = mk2(randr(F.fcx + o19, F.fcy + o19, frame), randr(F.fcy + o41, F.fcx + o41, frame));
    V3 N0 = normalize(hn0);
    V3 wi = sampleHemisphereCosine(u.pi, N0, uu);
    float cosTheta = fmaxr(dot(N0, wi), 0.0f);
    if (cosTheta <= MIN_COS_THETA) return mk3(0.0f);
    V3 origin = hp0 + N0 * u.eps;
    V3 hp1, hn1;
    int hit1 = tr.gi(origin, wi, hp1, hn1);
    if (hit1 < 0) return mk3(0.0f);
    V3 Li = (hit1 > 0) ? directLightBVH(tr, F, SEG_GI_DIRECT, hp1, hn1, frame, -wi) : sky<COUNT>(F, wi, w);
    V3 contrib = albedo0 * (cosTheta / u.pi) * Li;
    float lum = dot(contrib, mk3(0.299f, 0.587f, 0.114f));
    if (lum > MAX_GI_LUM) {
        float s = MAX_GI_LUM / fmaxr(lum, 1e-6f);
        contrib = contrib * s;
    }
    return contrib;
}

// computeAO (:721-757) in BVH mode:  bool T::ao(int i, V3 org, V3 dir, float radius) -> closest hit exists and t < radius
template <class T>
RT_DEV float computeAO_BVH(T &tr, const Frag &F, V3 hp, V3 hn, int frame) {
    const RtUniforms &u = *F.u;
    V3 N = normalize(hn);
    int occludedCount = 0;
    for (int i = 0; i < u.aoSamples; ++i) {
        float ox = (float)(37 * i + 3), oy = (float)(19 * i + 11);
        V2 uu = mk2(randr(F.fcx + ox, F.fcy + ox, frame), randr(F.fcy + oy, F.fcx + oy, frame));
        V3 dir = sampleHemisphereCosine(u.pi, N, uu);
        V3 org = hp + N * u.aoBias;
        if (tr.ao(i, org, dir, u.aoRadius)) occludedCount++;
    }
    float occ = (float)occludedCount / (float)u.aoSamples;
    float ao = 1.0f - occ;
    return clampr(mixr(u.aoMin, 1.0f, ao), u.aoMin, 1.0f);
}

// One BVH-mode sample of rt.frag:105-117 on a primary hit.
template <class T, bool COUNT>
RT_DEV V3 shadeSampleBVH(T &tr, const Frag &F, V3 hp, V3 hn, V3 V, int seed, float ao, Work &w) {
    const RtUniforms &u = *F.u;
    V3 radiance = directLightBVH(tr, F, SEG_DIRECT, hp, hn, seed, V);
    if (u.enableGI == 1) radiance = radiance + u.giScaleBVH * oneBounceGIBVH<T, COUNT>(tr, F, hp, hn, F.frameIndex, seed, w);
    if (u.enableAO == 1) radiance = radiance * ao;
    return radiance;
}

// ---------------------------------------------------------------------------------------------
// rt_taa.glsl:47-180.  History is addressed through `Hist`, which knows the tile-major layout:
//   V4 Hist::own()                 the pixel's own history texel (still branch, uv == vUV)
//   V4 Hist::at(float u, float v)  NEAREST + CLAMP_TO_EDGE fetch at an arbitrary uv (reprojection)
template <class Hist>
RT_DEV V4 resolveTAA(const RtUniforms &u, V3 curr, float uvx, float uvy, V2 motionOut, int frameIndex, Hist &hist) {
    const V3 Y = mk3(0.299f, 0.587f, 0.114f);
    float lCurr = dot(curr, Y);
    float lCurr2 = lCurr * lCurr;
    if (u.enableTAA == 0) return mk4(curr.x, curr.y, curr.z, lCurr2);
    if (frameIndex == 0) return mk4(curr.x, curr.y, curr.z, lCurr2);
    float motMag = length(motionOut);
    float MAX_W = u.taaHistoryMaxWeight, BOX = u.taaHistoryBoxSize;
    if (motMag < u.taaStillThresh) {
        V4 pr = hist.own();
        V3 prevCol = mk3(pr.x, pr.y, pr.z);
        float wHist = (frameIndex < 8) ? u.taaHistoryMinWeight : ((frameIndex < 32) ? u.taaHistoryAvgWeight : MAX_W);
        float wCurr = 1.0f - wHist;
        V3 meanNew = prevCol * wHist + curr * wCurr;
        float m2New = pr.w * wHist + lCurr2 * wCurr;
        return mk4(meanNew.x, meanNew.y, meanNew.z, m2New);
    }
    float upx = uvx - motionOut.x * 0.5f, upy = uvy - motionOut.y * 0.5f;
    bool oob = (upx < 0.0f || upy < 0.0f) || (upx > 1.0f || upy > 1.0f);
    if (oob) return mk4(curr.x, curr.y, curr.z, lCurr2);
    V4 pr = hist.at(upx, upy);
    V3 prevCol = mk3(pr.x, pr.y, pr.z);
    float wHist = 1.0f - smoothstepr(0.02f, u.taaHardMovingThresh, motMag);
    if (motMag > u.taaHardMovingThresh) wHist = 0.0f;
    float lPrev = dot(prevCol, Y);
    float maxL = fmaxr(fmaxr(lCurr, lPrev), 1e-3f);
    float relDiff = __builtin_fabsf(lCurr - lPrev) / maxL;
    float colorWeight = 1.0f - smoothstepr(0.03f, 0.25f, relDiff);
    wHist = wHist * colorWeight;
    bool bigColorChange = (motMag > 0.02f) && (relDiff > 0.30f);
    if (bigColorChange) wHist = 0.0f;
    wHist = clampr(wHist, 0.0f, MAX_W);
    float wCurr = 1.0f - wHist;
    V3 lo = curr - mk3(BOX), hi = curr + mk3(BOX);
    V3 hc = mk3(clampr(prevCol.x, lo.x, hi.x), clampr(prevCol.y, lo.y, hi.y), clampr(prevCol.z, lo.z, hi.z));
    V3 taaCol = wHist * hc + wCurr * curr;
    float m2New = wHist * pr.w + wCurr * lCurr2;
    return mk4(taaCol.x, taaCol.y, taaCol.z, m2New);
}

// Primary ray of rt.frag:58-68.  (jitterX, jitterY) = uJitter of the fragment's frame.
RT_DEV V3 primaryDirJ(const RtUniforms &u, float fcx, float fcy, float jitterX, float jitterY) {
    float jx = (u.enableJitter == 1) ? jitterX : 0.0f, jy = (u.enableJitter == 1) ? jitterY : 0.0f;
    float uvx = (fcx + jx) / u.resolution[0], uvy = (fcy + jy) / u.resolution[1];
    float nx = uvx * 2.0f - 1.0f, ny = uvy * 2.0f - 1.0f;
    V3 camRight = ld3(u.camRight), camUp = ld3(u.camUp), camFwd = ld3(u.camFwd);
    return normalize(camFwd + (nx * camRight) * (u.tanHalfFov * u.aspect) + (ny * camUp) * u.tanHalfFov);
}
RT_DEV V3 primaryDir(const RtUniforms &u, float fcx, float fcy) { return primaryDirJ(u, fcx, fcy, u.jitter[0], u.jitter[1]); }

}  // namespace rtd

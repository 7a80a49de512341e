// oracle/rt_oracle.cpp -- TEST INFRASTRUCTURE ONLY (parity oracle + CPU baseline). Never part of
// the product: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg load it.
//
// Scalar fp32 restatement of the reference's full-screen ray-trace fragment program, in the
// reference's include order (shaders/rt/rt.frag:41-47):
//   rt_common.glsl -> rt_materials.glsl -> rt_scene_analytic.glsl -> rt_bvh.glsl ->
//   rt_lighting.glsl -> rt_taa.glsl -> rt.frag main().
// GL semantics encoded here (none are written down in the reference): gl_FragCoord = pixel+0.5,
// row 0 = bottom row; NEAREST history fetch; non-seamless LINEAR cube map of u8 texels; RNE fp16
// on every MRT store (src/render/accum.cpp:10,25; src/render/gbuffer.cpp:52-53).
// Float model: oracle/orc_math.h.  Parity status: the reference has no tests or goldens; this
// file is pinned against the reference's own GLSL executed in the build container by SwiftShader (oracle/glsl_ref.py,
// fixtures tests/golden/glsl_*.npz, checked by tests/test_glsl_reference.py): analytic-scene frames, TAA, motion,
// materials, cube map, present pass, the BVH primitives (nodeFetch/triFetch/aabbHit/triHit) and -- since round 2 -- the
// traversal LOOPS of traceBVH/traceBVHShadow, ray by ray (2 816 rays, 552 of them equal-t ties that expose the visit
// order) and as whole BVH frames through rt.frag.  One load-time rewrite makes that possible (SwiftShader 4.1 mis-executes
// `continue` in that loop; `if (C) continue;` -> `if (!(C)) {...}`, oracle/glsl_ref.py structured_continue).  Not pinnable:
// what GLSL leaves to the driver (transcendentals, contraction, min/max of NaN), see DESIGN.md section 2.
#include <algorithm>
#include <atomic>
#include <cstdio>
#include <thread>
#include <vector>

#include "orc_math.h"
#include "orc_types.h"

namespace orc {

struct Counters {
    uint64_t raysClosest = 0, raysShadow = 0, raysAnalytic = 0, nodeFetch = 0, triFetch = 0, envLookup = 0,
             hitPixels = 0;
    // node+tri fetches split by the kind of ray that made them (the rest are the bounce's closest-hit rays)
    uint64_t fetchPrimary = 0, fetchShadow = 0, fetchAO = 0;
    uint64_t fetches() const { return nodeFetch + triFetch; }
};

// Everything a fragment invocation can see: uniforms + bound resources.
struct Scene {
    OrcUniforms u;
    const float *nodes = nullptr;   // 12 floats / node  (src/scene/bvh.cpp:147-168)
    const float *tris = nullptr;    // 12 floats / tri   (src/scene/bvh.cpp:187-204)
    const uint8_t *env = nullptr;   // 6 faces, +X -X +Y -Y +Z -Z, rows as uploaded (cubemap.cpp:86-91)
    int envSize = 0, envCh = 0;
    const uint16_t *prev = nullptr; // W*H*4 half, previous COLOR0
    int W = 0, H = 0;
    int envFilter = 0;              // cube-map filter model: 0 exact fp32 weights, 1 texel coordinates rounded to 1/256 texel
    int giBounces = 1;              // EXTENSION (not in the reference): diffuse bounces of the analytic / hybrid GI path, see giPath
};
// EXTENSION, not in the reference (SURVEY.md 8d config 3 "run B", labelled mode=hybrid): uUseBVH == 2 renders the analytic branch of
// rt.frag with the BVH mesh added to the analytic scene as one more object.  Everything else is the reference's analytic code.
static const int kSceneHybrid = 2;
// cube-map filter model (Scene::envFilter) for the orc_render / orc_texture_cube calls that follow.  0 = exact fp32 weights (default).
static std::atomic<int> g_envFilter{0};
static const int MAT_MESH = 5;      // no material of rt_materials.glsl:20-24 -> getMaterial's default branch (:123-124): 0.8 grey, spec 0.2, gloss 16, diffuse

// ---------------------------------------------------------------- rt_common.glsl
struct Hit { float t; vec3 p; vec3 n; int mat; };                       // :39-44

static inline uint32_t hash2(uint32_t vx, uint32_t vy) {                // :57-63
    vx = vx * 1664525u + 1013904223u;
    vy = vy * 1664525u + 1013904223u;
    vx ^= vy >> 16;
    vy ^= vx << 5;
    vx = vx * 1664525u + 1013904223u;
    vy = vy * 1664525u + 1013904223u;
    return vx ^ vy;
}
// uvec2(vec2): truncation toward zero; out-of-range is undefined in GLSL, pinned here to the clamp
// [0, 2^32-256] (what v_cvt_u32_f32 saturation gives on the ranges the shaders reach).
static inline uint32_t f2uint(float p) { return (uint32_t)fmin_(fmax_(p, 0.0f), 4294967040.0f); }
static inline uint32_t rand_bits(vec2 p, int frame) {                   // :75-77 (integer part)
    uint32_t fx = (uint32_t)frame, fy = (uint32_t)frame * 1663u;        // int32 wrap
    return hash2(f2uint(p.x) ^ fx, f2uint(p.y) ^ fy);
}
static inline float rand_(vec2 p, int frame) {                          // :75-77
    return (float)rand_bits(p, frame) / 4294967296.0f;
}
static inline float epsForDist(float d) { return fmax_(1e-4f, 1e-3f * d); }   // :88-90
static inline float halton(int i, int b) {                              // :106-116
    float f = 1.0f, r = 0.0f;
    int n = i;
    while (n > 0) {
        f /= (float)b;
        r += f * (float)(n % b);
        n /= b;
    }
    return r;
}
static inline vec2 ld2(int i) { return {halton(i + 1, 2), halton(i + 1, 3)}; }   // :127-129
static inline vec2 concentricSample(const Scene &S, vec2 u) {           // :144-159
    float a = 2.0f * u.x - 1.0f;
    float b = 2.0f * u.y - 1.0f;
    float r, phi;
    if (a == 0.0f && b == 0.0f) { r = 0.0f; phi = 0.0f; }
    else if (std::fabs(a) > std::fabs(b)) { r = a; phi = (S.u.pi / 4.0f) * (b / a); }
    else { r = b; phi = (S.u.pi / 2.0f) - (S.u.pi / 4.0f) * (a / b); }
    float s, c;
    sincos_core(phi, &s, &c);
    return {r * c, r * s};
}
static inline vec2 ndcFromWorld(vec3 p, const float *VP) {              // :175-179, column-major
    float cx = std::fmaf(VP[8], p.z, std::fmaf(VP[4], p.y, VP[0] * p.x)) + VP[12];
    float cy = std::fmaf(VP[9], p.z, std::fmaf(VP[5], p.y, VP[1] * p.x)) + VP[13];
    float cw = std::fmaf(VP[11], p.z, std::fmaf(VP[7], p.y, VP[3] * p.x)) + VP[15];
    float w = fmax_(cw, 1e-6f);
    return {cx / w, cy / w};
}

// ---------------------------------------------------------------- rt_materials.glsl
enum { MAT_FLOOR = 0, MAT_ALBEDO_SPHERE = 1, MAT_GLASS_SPHERE = 2, MAT_MIRROR_SPHERE = 3, MAT_POINTLIGHT_SPHERE = 4 };
struct MaterialProps { vec3 albedo; float specStrength; float gloss; int type; float ior; };   // :36-42

static inline vec3 ld3(const float *p) { return {p[0], p[1], p[2]}; }

static MaterialProps getMaterial(const Scene &S, int id) {             // :57-125
    const OrcUniforms &u = S.u;
    MaterialProps m;
    MaterialProps alb{ld3(u.matAlbedoColor), u.matAlbedoSpecStrength, u.matAlbedoGloss, 0, 1.0f};
    if (id == MAT_FLOOR) { m = {v3(0.7f), 0.1f, 16.0f, 0, 1.0f}; return m; }
    if (id == MAT_ALBEDO_SPHERE) return alb;
    if (id == MAT_GLASS_SPHERE) {
        if (u.matGlassEnabled == 0) return alb;
        m = {ld3(u.matGlassAlbedo), u.matGlassDistortion, 1.0f, 2, u.matGlassIOR};
        return m;
    }
    if (id == MAT_MIRROR_SPHERE) {
        if (u.matMirrorEnabled == 0) return alb;
        m = {ld3(u.matMirrorAlbedo), 0.0f, u.matMirrorGloss, 1, 1.0f};
        return m;
    }
    m = {v3(0.8f), 0.2f, 16.0f, 0, 1.0f};
    return m;
}

// ---------------------------------------------------------------- rt_scene_analytic.glsl
static const vec3 kFloorNormal = {0.0f, 1.0f, 0.0f};                    // :37-54
static const float kFloorD = 0.0f;
static const vec3 kSphereLeftCenter = {-1.2f, 1.0f, -3.5f};
static const float kSphereLeftRadius = 1.0f;
static const vec3 kGlassCenter = {0.7f, 1.0f, -5.0f};
static const float kGlassRadius = 1.0f;
static const vec3 kMirrorCenter = {1.2f, 0.7f, -2.5f};
static const float kMirrorRadius = 0.7f;
static const float kPointLightRadius = 0.15f;

static bool intersectPlane(const Scene &S, vec3 ro, vec3 rd, vec3 n, float d, Hit &h, int matId) {   // :71-81
    float denom = dot(n, rd);
    if (std::fabs(denom) < 1e-6f) return false;
    float t = -(dot(n, ro) + d) / denom;
    if (t < S.u.eps) return false;
    h.t = t;
    h.p = ro + rd * t;
    h.n = n;
    h.mat = matId;
    return true;
}
static bool intersectSphere(const Scene &S, vec3 ro, vec3 rd, vec3 c, float r, Hit &h, int matId) {  // :96-111
    vec3 oc = ro - c;
    float b = dot(oc, rd);
    float c2 = dot(oc, oc) - r * r;
    float disc = b * b - c2;
    if (disc < 0.0f) return false;
    float s = std::sqrt(disc);
    float t = -b - s;
    if (t < S.u.eps) t = -b + s;
    if (t < S.u.eps) return false;
    h.t = t;
    h.p = ro + rd * t;
    h.n = normalize(h.p - c);
    h.mat = matId;
    return true;
}
static bool traceBVH(const Scene &S, Counters &C, vec3 ro, vec3 rd, Hit &hitOut);
static bool traceAnalyticCore(const Scene &S, Counters &C, vec3 ro, vec3 rd, bool includeGlass,
                              bool includePointLightSphere, Hit &hit) {                              // :132-167
    C.raysAnalytic++;
    hit.t = S.u.inf;
    Hit h;
    if (intersectPlane(S, ro, rd, kFloorNormal, kFloorD, h, MAT_FLOOR) && h.t < hit.t) hit = h;
    if (intersectSphere(S, ro, rd, kSphereLeftCenter, kSphereLeftRadius, h, MAT_ALBEDO_SPHERE) && h.t < hit.t) hit = h;
    if (includeGlass) {
        if (intersectSphere(S, ro, rd, kGlassCenter, kGlassRadius, h, MAT_GLASS_SPHERE) && h.t < hit.t) hit = h;
    }
    if (intersectSphere(S, ro, rd, kMirrorCenter, kMirrorRadius, h, MAT_MIRROR_SPHERE) && h.t < hit.t) hit = h;
    if (includePointLightSphere && S.u.pointLightEnabled == 1) {
        if (intersectSphere(S, ro, rd, ld3(S.u.pointLightPos), kPointLightRadius, h, MAT_POINTLIGHT_SPHERE) && h.t < hit.t)
            hit = h;
    }
    if (S.u.useBVH == kSceneHybrid) {
        // EXTENSION: the mesh is the last object of the list, under the list's own rule (strict <: an earlier object wins a tie);
        // its hit is what traceBVH returns (geometric normal), its material id the mesh's
        if (traceBVH(S, C, ro, rd, h) && h.t < hit.t) { hit = h; hit.mat = MAT_MESH; }
    }
    return hit.t < S.u.inf;
}
static bool traceAnalytic(const Scene &S, Counters &C, vec3 ro, vec3 rd, Hit &h) { return traceAnalyticCore(S, C, ro, rd, true, true, h); }               // :175
static bool traceAnalyticIgnoreGlass(const Scene &S, Counters &C, vec3 ro, vec3 rd, Hit &h) { return traceAnalyticCore(S, C, ro, rd, false, true, h); }   // :185
static bool traceAnalyticIgnorePointLight(const Scene &S, Counters &C, vec3 ro, vec3 rd, Hit &h) { return traceAnalyticCore(S, C, ro, rd, true, false, h); }  // :195

// texture(samplerCube, dir): OpenGL 4.1 core 3.8.10 face selection, LINEAR within the face,
// CLAMP_TO_EDGE, no seamless filtering (src/render/cubemap.cpp:96-100; no glEnable(SEAMLESS)).
static vec3 textureCube(const Scene &S, Counters &C, vec3 d) {
    C.envLookup++;
    float ax = std::fabs(d.x), ay = std::fabs(d.y), az = std::fabs(d.z);
    int face; float sc, tc, ma;
    if (ax >= ay && ax >= az) { ma = ax; if (d.x >= 0.0f) { face = 0; sc = -d.z; tc = -d.y; } else { face = 1; sc = d.z; tc = -d.y; } }
    else if (ay >= az)        { ma = ay; if (d.y >= 0.0f) { face = 2; sc = d.x; tc = d.z; } else { face = 3; sc = d.x; tc = -d.z; } }
    else                      { ma = az; if (d.z >= 0.0f) { face = 4; sc = d.x; tc = -d.y; } else { face = 5; sc = -d.x; tc = -d.y; } }
    float s = 0.5f * (sc / ma + 1.0f);
    float t = 0.5f * (tc / ma + 1.0f);
    const int N = S.envSize, ch = S.envCh;
    float fu = s * (float)N - 0.5f, fv = t * (float)N - 0.5f;
    if (S.envFilter == 1) {   // SURVEY.md 8c's second mode: GPU samplers filter RGB8 with ~8 bits of sub-texel precision -- coordinates rounded
        fu = std::floor(fu * 256.0f + 0.5f) * 0.00390625f;   // to nearest on a 1/256-texel grid; everything after is unchanged
        fv = std::floor(fv * 256.0f + 0.5f) * 0.00390625f;
    }
    float flu = std::floor(fu), flv = std::floor(fv);
    float a = fu - flu, b = fv - flv;
    int i0 = (int)flu, j0 = (int)flv;
    int i1 = i0 + 1, j1 = j0 + 1;
    i0 = std::min(std::max(i0, 0), N - 1); i1 = std::min(std::max(i1, 0), N - 1);
    j0 = std::min(std::max(j0, 0), N - 1); j1 = std::min(std::max(j1, 0), N - 1);
    const uint8_t *F = S.env + (size_t)face * N * N * ch;
    auto tex = [&](int i, int j) {
        const uint8_t *p = F + ((size_t)j * N + i) * ch;
        return v3((float)p[0] / 255.0f, (float)p[1] / 255.0f, (float)p[2] / 255.0f);
    };
    vec3 t00 = tex(i0, j0), t10 = tex(i1, j0), t01 = tex(i0, j1), t11 = tex(i1, j1);
    float w00 = (1.0f - a) * (1.0f - b), w10 = a * (1.0f - b), w01 = (1.0f - a) * b, w11 = a * b;
    return t00 * w00 + t10 * w10 + t01 * w01 + t11 * w11;
}
static vec3 sky(const Scene &S, Counters &C, vec3 dir) {               // :211-223
    if (S.u.useEnvMap == 1) return textureCube(S, C, dir) * S.u.envIntensity;
    float t = clampf(0.5f * (dir.y + 1.0f), 0.0f, 1.0f);
    return mix(v3(0.6f, 0.7f, 0.9f) * 0.3f, v3(0.1f, 0.15f, 0.3f) * 0.3f, 1.0f - t);
}

// ---------------------------------------------------------------- rt_bvh.glsl
struct TriSOA { vec3 v0, e1, e2; };                                      // :38-42
struct NodeSOA { vec3 bmin; int left; vec3 bmax; int right; int first; int count; };   // :77-81

static inline TriSOA triFetch(const Scene &S, Counters &C, int i) {     // :55-65
    C.triFetch++;
    const float *p = S.tris + (size_t)i * 12;
    return {{p[0], p[1], p[2]}, {p[4], p[5], p[6]}, {p[8], p[9], p[10]}};
}
static inline NodeSOA nodeFetch(const Scene &S, Counters &C, int i) {   // :91-102
    C.nodeFetch++;
    const float *p = S.nodes + (size_t)i * 12;
    NodeSOA N;
    N.bmin = {p[0], p[1], p[2]}; N.left = (int)(p[3] + 0.5f);
    N.bmax = {p[4], p[5], p[6]}; N.right = (int)(p[7] + 0.5f);
    N.first = (int)(p[8] + 0.5f);
    N.count = (int)(p[9] + 0.5f);
    return N;
}
static inline bool aabbHit(vec3 ro, vec3 rdInv, vec3 bmin, vec3 bmax, float &tminOut, float &tmaxOut) {   // :124-134
    vec3 t0 = (bmin - ro) * rdInv;
    vec3 t1 = (bmax - ro) * rdInv;
    vec3 tsm = {fmin_(t0.x, t1.x), fmin_(t0.y, t1.y), fmin_(t0.z, t1.z)};
    vec3 tbg = {fmax_(t0.x, t1.x), fmax_(t0.y, t1.y), fmax_(t0.z, t1.z)};
    float tmin = fmax_(fmax_(tsm.x, tsm.y), fmax_(tsm.z, 0.0f));
    float tmax = fmin_(fmin_(tbg.x, tbg.y), tbg.z);
    tminOut = tmin;
    tmaxOut = tmax;
    return tmax >= tmin;
}
static inline bool triHit(const Scene &S, vec3 ro, vec3 rd, const TriSOA &T, float tMax, float &t, vec3 &n) {   // :154-170
    vec3 pvec = cross(rd, T.e2);
    float det = dot(T.e1, pvec);
    if (std::fabs(det) < 1e-8f) return false;
    float invDet = 1.0f / det;
    vec3 tvec = ro - T.v0;
    float u = dot(tvec, pvec) * invDet;
    if (u < 0.0f || u > 1.0f) return false;
    vec3 qvec = cross(tvec, T.e1);
    float v = dot(rd, qvec) * invDet;
    if (v < 0.0f || u + v > 1.0f) return false;
    float tt = dot(T.e2, qvec) * invDet;
    if (tt < S.u.eps || tt > tMax) return false;
    t = tt;
    n = normalize(cross(T.e1, T.e2));
    return true;
}
static bool traceBVH(const Scene &S, Counters &C, vec3 ro, vec3 rd, Hit &hitOut) {    // :193-243
    C.raysClosest++;
    if (S.u.nodeCount <= 0 || S.u.triCount <= 0) return false;
    hitOut.t = S.u.inf;
    hitOut.n = v3(0.0f);
    hitOut.mat = 1;
    float tminBox, tmaxBox;
    vec3 rdInv = {1.0f / rd.x, 1.0f / rd.y, 1.0f / rd.z};
    int stack[64];
    int sp = 0;
    stack[sp++] = 0;
    while (sp > 0) {
        int ni = stack[--sp];
        NodeSOA N = nodeFetch(S, C, ni);
        if (!aabbHit(ro, rdInv, N.bmin, N.bmax, tminBox, tmaxBox) || tminBox > hitOut.t) continue;
        if (N.count > 0) {
            for (int i = 0; i < N.count; ++i) {
                TriSOA T = triFetch(S, C, N.first + i);
                float t; vec3 n;
                if (triHit(S, ro, rd, T, hitOut.t, t, n)) {
                    hitOut.t = t;
                    hitOut.p = ro + rd * t;
                    hitOut.n = n;
                    hitOut.mat = 1;
                }
            }
        } else {
            NodeSOA L = nodeFetch(S, C, N.left);
            NodeSOA R = nodeFetch(S, C, N.right);
            float tminL, tmaxL, tminR, tmaxR;
            bool hitL = aabbHit(ro, rdInv, L.bmin, L.bmax, tminL, tmaxL) && tminL <= hitOut.t;
            bool hitR = aabbHit(ro, rdInv, R.bmin, R.bmax, tminR, tmaxR) && tminR <= hitOut.t;
            if (hitL && hitR) {
                bool leftFirst = tminL < tminR;
                stack[sp++] = leftFirst ? N.right : N.left;
                stack[sp++] = leftFirst ? N.left : N.right;
            } else if (hitL) stack[sp++] = N.left;
            else if (hitR) stack[sp++] = N.right;
        }
    }
    return hitOut.t < S.u.inf;
}
static bool traceBVHShadowImpl(const Scene &S, Counters &C, vec3 ro, vec3 rd, float tMax);
static bool traceBVHShadow(const Scene &S, Counters &C, vec3 ro, vec3 rd, float tMax) {
    uint64_t f0 = C.fetches();
    bool r = traceBVHShadowImpl(S, C, ro, rd, tMax);
    C.fetchShadow += C.fetches() - f0;
    return r;
}
static bool traceBVHShadowImpl(const Scene &S, Counters &C, vec3 ro, vec3 rd, float tMax) {   // :260-304
    C.raysShadow++;
    if (S.u.nodeCount <= 0 || S.u.triCount <= 0) return false;
    float tminBox, tmaxBox;
    vec3 rdInv = {1.0f / rd.x, 1.0f / rd.y, 1.0f / rd.z};
    int stack[64];
    int sp = 0;
    stack[sp++] = 0;
    while (sp > 0) {
        int ni = stack[--sp];
        NodeSOA N = nodeFetch(S, C, ni);
        if (!aabbHit(ro, rdInv, N.bmin, N.bmax, tminBox, tmaxBox) || tminBox > tMax) continue;
        if (N.count > 0) {
            for (int i = 0; i < N.count; ++i) {
                TriSOA T = triFetch(S, C, N.first + i);
                float t; vec3 n;
                if (triHit(S, ro, rd, T, tMax, t, n)) return true;
            }
        } else {
            NodeSOA L = nodeFetch(S, C, N.left);
            NodeSOA R = nodeFetch(S, C, N.right);
            float tminL, tmaxL, tminR, tmaxR;
            bool hitL = aabbHit(ro, rdInv, L.bmin, L.bmax, tminL, tmaxL) && tminL <= tMax;
            bool hitR = aabbHit(ro, rdInv, R.bmin, R.bmax, tminR, tmaxR) && tminR <= tMax;
            if (hitL && hitR) {
                bool leftFirst = tminL < tminR;
                stack[sp++] = leftFirst ? N.right : N.left;
                stack[sp++] = leftFirst ? N.left : N.right;
            } else if (hitL) stack[sp++] = N.left;
            else if (hitR) stack[sp++] = N.right;
        }
    }
    return false;
}

// ---------------------------------------------------------------- rt_lighting.glsl
static const vec3 kLightCenter = {0.0f, 5.0f, -3.0f};                   // :29-32
static const float kLightRadius = 1.2f;
static const vec3 kLightCol = {18.0f, 18.0f, 18.0f};
static inline vec3 kLightN() { return normalize(v3(0.0f, -1.0f, 0.2f)); }

// Per-fragment context: gl_FragCoord.
struct Frag { vec2 fc; };

static bool occludedToward(const Scene &S, Counters &C, vec3 p, vec3 q) {    // :49-60
    vec3 rd = normalize(q - p);
    float maxT = length(q - p);
    float eps = epsForDist(maxT);
    if (S.u.useBVH == 1) return traceBVHShadow(S, C, p + rd * eps, rd, maxT - eps);
    Hit h;
    if (traceAnalytic(S, C, p + rd * eps, rd, h) && h.t < maxT - eps) return true;
    return false;
}
static vec3 shadeLambertPhong(const Scene &S, vec3 N, vec3 V, vec3 L, vec3 Li, vec3 albedo, float specStrength, float gloss) {   // :78-98
    float ndl = fmax_(dot(N, L), 0.0f);
    if (ndl <= 0.0f) return v3(0.0f);
    vec3 diffuse = albedo * (ndl / S.u.pi);
    vec3 spec = v3(0.0f);
    if (specStrength > 0.0f) {
        vec3 H = normalize(L + V);
        float ndh = fmax_(dot(N, H), 0.0f);
        float phong = powf_(ndh, gloss);
        spec = (specStrength * phong) * v3(1.0f);
    }
    return (diffuse + spec) * Li;
}
static vec3 sunDirect(const Scene &S, Counters &C, const Hit &h, const MaterialProps &mat, vec3 Vdir) {   // :114-144
    if (S.u.sunEnabled == 0) return v3(0.0f);
    vec3 N = normalize(h.n);
    vec3 V = normalize(Vdir);
    vec3 L = normalize(-ld3(S.u.sunDir));
    float ndl = fmax_(dot(N, L), 0.0f);
    if (ndl <= 0.0f) return v3(0.0f);
    float maxT = 1000.0f;
    float eps = epsForDist(maxT);
    vec3 origin = h.p + N * eps;
    bool blocked;
    if (S.u.useBVH == 1) blocked = traceBVHShadow(S, C, origin, L, maxT - eps);
    else { Hit tmp; blocked = traceAnalytic(S, C, origin, L, tmp); }
    if (blocked) return v3(0.0f);
    vec3 Li = ld3(S.u.sunColor) * S.u.sunIntensity;
    float specStrength = (mat.type == 0) ? mat.specStrength : 0.0f;
    return shadeLambertPhong(S, N, V, L, Li, mat.albedo, specStrength, mat.gloss);
}
static vec3 skyDirect(const Scene &S, const Hit &h, const MaterialProps &mat) {   // :156-169
    if (S.u.skyEnabled == 0) return v3(0.0f);
    vec3 N = normalize(h.n);
    vec3 U = normalize(ld3(S.u.skyUpDir));
    float ndl = fmax_(dot(N, U), 0.0f);
    if (ndl <= 0.0f) return v3(0.0f);
    vec3 Li = ld3(S.u.skyColor) * S.u.skyIntensity;
    return mat.albedo * (ndl / S.u.pi) * Li;
}
static vec3 pointDirect(const Scene &S, Counters &C, const Hit &h, const MaterialProps &mat, vec3 Vdir) {   // :181-214
    if (S.u.pointLightEnabled == 0) return v3(0.0f);
    vec3 N = normalize(h.n);
    vec3 V = normalize(Vdir);
    vec3 toL = ld3(S.u.pointLightPos) - h.p;
    float dist2 = dot(toL, toL);
    if (dist2 <= 1e-6f) return v3(0.0f);
    float dist = std::sqrt(dist2);
    vec3 L = toL / dist;
    float ndl = fmax_(dot(N, L), 0.0f);
    if (ndl <= 0.0f) return v3(0.0f);
    float eps = epsForDist(dist);
    vec3 origin = h.p + L * eps;
    bool blocked;
    if (S.u.useBVH == 1) blocked = traceBVHShadow(S, C, origin, L, dist - eps);
    else { Hit tmp; blocked = traceAnalyticIgnorePointLight(S, C, origin, L, tmp) && tmp.t < dist - eps; }
    if (blocked) return v3(0.0f);
    vec3 Li = ld3(S.u.pointLightColor) * (S.u.pointLightIntensity / fmax_(dist2, 1e-4f));
    float specStrength = (mat.type == 0) ? mat.specStrength : 0.0f;
    return shadeLambertPhong(S, N, V, L, Li, mat.albedo, specStrength, mat.gloss);
}
static void buildONB(vec3 N, vec3 &T, vec3 &B) {                       // :227-231
    vec3 up = (std::fabs(N.y) < 0.99f) ? v3(0.0f, 1.0f, 0.0f) : v3(1.0f, 0.0f, 0.0f);
    T = normalize(cross(up, N));
    B = cross(N, T);
}
static vec3 sampleHemisphereCosine(const Scene &S, vec3 N, vec2 u) {   // :251-266
    float phi = 2.0f * S.u.pi * u.x;
    float r = std::sqrt(u.y);
    float sn, cs;
    sincos_core(phi, &sn, &cs);
    float x = r * cs;
    float z = r * sn;
    float y = std::sqrt(fmax_(0.0f, 1.0f - u.y));
    vec3 T, B;
    buildONB(normalize(N), T, B);
    return normalize(x * T + z * B + y * N);
}
static vec2 cpOffset(vec2 pix, int frame) {                            // :280-289
    vec2 h = {rand_(pix, (int)((uint32_t)frame * 911u)), rand_(vec2{pix.y, pix.x}, (int)((uint32_t)frame * 577u))};
    vec2 ld = ld2(frame);
    return {fractf(h.x + ld.x), fractf(h.y + ld.y)};
}
// Disk-light tangent frame, rt_lighting.glsl:355-357 / :414-416.
static void lightFrame(vec3 &t, vec3 &b) {
    vec3 n = kLightN();
    t = normalize(std::fabs(n.y) < 0.99f ? cross(n, v3(0.0f, 1.0f, 0.0f)) : cross(n, v3(1.0f, 0.0f, 0.0f)));
    b = cross(n, t);
}
// The 4-sample disk loop shared by directLight (:363-387) and directLightBVH (:422-445).
static vec3 diskLight(const Scene &S, Counters &C, const Frag &F, const Hit &h, vec3 N, vec3 V, int frame, vec3 albedo,
                      float specStrength, float gloss) {
    vec3 t, b;
    lightFrame(t, b);
    vec2 rot = cpOffset(F.fc, S.u.frameIndex);
    vec3 sum = v3(0.0f);
    vec3 lightN = kLightN();
    for (int i = 0; i < 4; ++i) {                                      // SOFT_SHADOW_SAMPLES rt_common.glsl:23
        vec2 u = {rand_(vec2{F.fc.x + (float)i, F.fc.y + (float)i}, frame),
                  rand_(vec2{F.fc.y + (float)(31 * i + 7), F.fc.x + (float)(31 * i + 7)}, frame)};
        u = {fractf(u.x + rot.x), fractf(u.y + rot.y)};
        vec2 cd = concentricSample(S, u);
        vec2 d = {cd.x * kLightRadius, cd.y * kLightRadius};
        vec3 xL = kLightCenter + t * d.x + b * d.y;
        vec3 L = normalize(xL - h.p);
        float ndl = fmax_(dot(N, L), 0.0f);
        float cosThetaL = fmax_(dot(-lightN, L), 0.0f);
        float r2 = fmax_(dot(xL - h.p, xL - h.p), 1e-4f);
        float geom = (ndl * cosThetaL) / r2;
        float vis = occludedToward(S, C, h.p, xL) ? 0.0f : 1.0f;
        vec3 Li = kLightCol * geom * vis;
        sum += shadeLambertPhong(S, N, V, L, Li, albedo, specStrength, gloss);
    }
    return sum / 4.0f;
}
static vec3 directLight(const Scene &S, Counters &C, const Frag &F, const Hit &h, int frame, vec3 Vdir) {   // :313-395
    vec3 N = normalize(h.n);
    MaterialProps mat = getMaterial(S, h.mat);
    vec3 V = normalize(Vdir);
    if (mat.type == 1) {
        vec3 R = reflect(-V, N);
        vec3 col = (S.u.useEnvMap == 1) ? textureCube(S, C, R) * S.u.envIntensity : sky(S, C, R);
        return col * mat.albedo;
    }
    if (mat.type == 2) {
        vec3 R = reflect(-V, N);
        vec3 refl = (S.u.useEnvMap == 1) ? textureCube(S, C, R) * S.u.envIntensity : sky(S, C, R);
        vec3 skyDiff = skyDirect(S, h, mat);
        return refl * mat.albedo + skyDiff;
    }
    vec3 sum = diskLight(S, C, F, h, N, V, frame, mat.albedo, mat.specStrength, mat.gloss);
    sum += sunDirect(S, C, h, mat, V);
    sum += skyDirect(S, h, mat);
    sum += pointDirect(S, C, h, mat, V);
    return sum;
}
static vec3 directLightBVH(const Scene &S, Counters &C, const Frag &F, const Hit &h, int frame, vec3 Vdir) {   // :405-460
    vec3 N = normalize(h.n);
    const vec3 albedo = v3(0.85f);
    const float specStrength = 0.25f, gloss = 32.0f;
    vec3 V = normalize(Vdir);
    vec3 sum = diskLight(S, C, F, h, N, V, frame, albedo, specStrength, gloss);
    MaterialProps fakeMat{albedo, specStrength, gloss, 0, 1.0f};
    sum += sunDirect(S, C, h, fakeMat, V);
    sum += skyDirect(S, h, fakeMat);
    sum += pointDirect(S, C, h, fakeMat, V);
    return sum;
}
// oneBounceGIAnalytic, rt_lighting.glsl:473-507, generalised to S.giBounces diffuse bounces (EXTENSION; with giBounces == 1 --
// the default and the only thing the reference does -- this is that function, operation for operation: 1 * x and 0 + x are exact).
// Level k = 0, 1, ... starts at hit h_k with seed_k (seed_{k+1} = seed_k * 131 + 17, the derivation shadeMirror uses for its nested
// GI, :692) and throughput T_k (T_0 = 1):
//   F = albedo(h_k) * (cos / pi);  Li = directLight(h_{k+1}) if the bounce ray hits, else sky(wi) and the path ends;
//   result += (T_k * F) * Li;   T_{k+1} = (T_k * F) * giScaleAnalytic.
// The device code (csrc/rt_device_analytic.hpp) performs the same operations in the same order.
static const int kMaxGiBounces = 8;
static vec3 oneBounceGIAnalytic(const Scene &S, Counters &C, const Frag &F, const Hit &h0, int frame, int seed) {
    vec3 result = v3(0.0f), T = v3(1.0f);
    Hit h = h0;
    const int maxB = S.giBounces < 1 ? 1 : (S.giBounces > kMaxGiBounces ? kMaxGiBounces : S.giBounces);
    for (int k = 0; k < maxB; ++k) {
        MaterialProps mat0 = getMaterial(S, h.mat);
        vec3 N0 = normalize(h.n);
        float o13 = (float)(int)((uint32_t)seed * 13u), o37 = (float)(int)((uint32_t)seed * 37u);
        vec2 u = {rand_(vec2{F.fc.x + o13, F.fc.y + o13}, frame), rand_(vec2{F.fc.y + o37, F.fc.x + o37}, frame)};
        vec3 wi = sampleHemisphereCosine(S, N0, u);
        float cosTheta = fmax_(dot(N0, wi), 0.0f);
        if (cosTheta <= 0.0f) break;
        vec3 origin = h.p + N0 * S.u.eps;
        Hit h1;
        bool hit1 = traceAnalytic(S, C, origin, wi, h1);
        vec3 Li = hit1 ? directLight(S, C, F, h1, frame, -wi) : sky(S, C, wi);
        vec3 TF = T * (mat0.albedo * (cosTheta / S.u.pi));
        result = result + TF * Li;
        if (!hit1) break;
        T = TF * S.u.giScaleAnalytic;
        h = h1;
        seed = (int)((uint32_t)seed * 131u + 17u);
    }
    return result;
}
static vec3 oneBounceGIBVH(const Scene &S, Counters &C, const Frag &F, const Hit &h0, int frame, int seed) {   // :515-561
    const vec3 albedo0 = v3(0.85f);
    const float MAX_GI_LUM = 8.0f, MIN_COS_THETA = 0.1f;
    float o19 = (float)(int)((uint32_t)seed * 19u), o41 = (float)(int)((uint32_t)seed * 41u);
    vec2 u = {rand_(vec2{F.fc.x + o19, F.fc.y + o19}, frame), rand_(vec2{F.fc.y + o41, F.fc.x + o41}, frame)};
    vec3 N0 = normalize(h0.n);
    vec3 wi = sampleHemisphereCosine(S, N0, u);
    float cosTheta = fmax_(dot(N0, wi), 0.0f);
    if (cosTheta <= MIN_COS_THETA) return v3(0.0f);
    vec3 origin = h0.p + N0 * S.u.eps;
    Hit h1;
    bool hit1 = traceBVH(S, C, origin, wi, h1);
    vec3 Li = hit1 ? directLightBVH(S, C, F, h1, frame, -wi) : sky(S, C, wi);
    vec3 contrib = albedo0 * (cosTheta / S.u.pi) * Li;
    float lum = dot(contrib, v3(0.299f, 0.587f, 0.114f));
    if (lum > MAX_GI_LUM) {
        float s = MAX_GI_LUM / fmax_(lum, 1e-6f);
        contrib *= s;
    }
    return contrib;
}
static vec3 shadeGlass(const Scene &S, Counters &C, const Frag &F, const Hit &h, vec3 wo, const MaterialProps &mat, int frame) {   // :576-663
    vec3 N = normalize(h.n);
    vec3 V = normalize(wo);
    vec3 I = -V;
    float ior = mat.ior;
    float eta = 1.0f / fmax_(ior, 1.0001f);
    const float distortionStrength = 0.45f;
    vec3 camPos = ld3(S.u.camPos);
    vec3 R = reflect(I, N);
    vec3 reflectEnv = sky(S, C, R);
    vec3 reflectLocal = reflectEnv;
    {
        Hit hRefl;
        if (traceAnalyticIgnoreGlass(S, C, h.p + R * S.u.eps, R, hRefl)) {
            vec3 V2 = normalize(camPos - hRefl.p);
            reflectLocal = directLight(S, C, F, hRefl, frame, V2);
        }
    }
    const float localReflWeight = 0.4f;
    vec3 reflectCol = mix(reflectEnv, reflectLocal, localReflWeight);
    vec3 straightCol;
    {
        Hit hStraight;
        if (traceAnalyticIgnoreGlass(S, C, h.p + I * S.u.eps, I, hStraight)) {
            vec3 V2 = normalize(camPos - hStraight.p);
            straightCol = directLight(S, C, F, hStraight, frame, V2);
        } else straightCol = sky(S, C, I);
    }
    float cosTheta = clampf(dot(-I, N), 0.0f, 1.0f);
    float k = 1.0f - eta * eta * (1.0f - cosTheta * cosTheta);
    vec3 refrCol = straightCol;
    if (distortionStrength > 0.0f && k > 0.0f) {
        vec3 T_phys = normalize(refract(I, N, eta));
        vec3 T = normalize(mix(I, T_phys, distortionStrength));
        Hit hRefr;
        vec3 bentCol;
        if (traceAnalyticIgnoreGlass(S, C, h.p + T * S.u.eps, T, hRefr)) {
            vec3 V2 = normalize(camPos - hRefr.p);
            bentCol = directLight(S, C, F, hRefr, frame, V2);
        } else bentCol = sky(S, C, T);
        refrCol = mix(straightCol, bentCol, distortionStrength);
    }
    refrCol *= mat.albedo;
    float F0 = powf_((ior - 1.0f) / (ior + 1.0f), 2.0f);
    float fresnel = F0 + (1.0f - F0) * powf_(1.0f - cosTheta, 5.0f);
    return mix(refrCol, reflectCol, fresnel);
}
static vec3 shadeMirror(const Scene &S, Counters &C, const Frag &F, const Hit &h, vec3 wo, const MaterialProps &mat, int frame) {   // :675-708
    vec3 N = normalize(h.n);
    vec3 I = -normalize(wo);
    vec3 R = reflect(I, N);
    vec3 org = h.p + R * S.u.eps;
    Hit h2;
    bool hit2 = traceAnalytic(S, C, org, R, h2);
    vec3 col;
    if (hit2) {
        col = directLight(S, C, F, h2, frame, -R);
        if (S.u.enableGI == 1) {
            int giSeed = (int)((uint32_t)frame * 131u + 17u);
            col += S.u.giScaleAnalytic * oneBounceGIAnalytic(S, C, F, h2, frame, giSeed);
        }
    } else {
        col = (S.u.useEnvMap == 1) ? textureCube(S, C, R) * S.u.envIntensity : sky(S, C, R);
    }
    col *= mat.albedo;
    return col;
}
static float computeAO(const Scene &S, Counters &C, const Frag &F, const Hit &h, int frame) {   // :721-757
    vec3 N = normalize(h.n);
    int occludedCount = 0;
    for (int i = 0; i < S.u.aoSamples; ++i) {
        float ox = (float)(37 * i + 3), oy = (float)(19 * i + 11);
        vec2 u = {rand_(vec2{F.fc.x + ox, F.fc.y + ox}, frame), rand_(vec2{F.fc.y + oy, F.fc.x + oy}, frame)};
        vec3 dir = sampleHemisphereCosine(S, N, u);
        vec3 org = h.p + N * S.u.aoBias;
        Hit tmp;
        uint64_t f0 = C.fetches();
        bool hitAny = (S.u.useBVH == 1) ? traceBVH(S, C, org, dir, tmp) : traceAnalytic(S, C, org, dir, tmp);
        C.fetchAO += C.fetches() - f0;
        if (hitAny && tmp.t < S.u.aoRadius) occludedCount++;
    }
    float occ = (float)occludedCount / (float)S.u.aoSamples;
    float ao = 1.0f - occ;
    ao = clampf(mixf(S.u.aoMin, 1.0f, ao), S.u.aoMin, 1.0f);
    return ao;
}

// ---------------------------------------------------------------- rt_taa.glsl
struct vec4 { float x, y, z, w; };
static vec4 fetchPrev(const Scene &S, float u, float v) {    // texture(prevAccum, uv), NEAREST + CLAMP_TO_EDGE
    int x = (int)std::floor(u * (float)S.W), y = (int)std::floor(v * (float)S.H);
    x = std::min(std::max(x, 0), S.W - 1);
    y = std::min(std::max(y, 0), S.H - 1);
    if (!S.prev) return {0, 0, 0, 0};
    const uint16_t *p = S.prev + ((size_t)y * S.W + x) * 4;
    return {f16_to_f32(p[0]), f16_to_f32(p[1]), f16_to_f32(p[2]), f16_to_f32(p[3])};
}
static vec4 resolveTAA(const Scene &S, vec3 curr, vec2 uvCurr, vec2 motionOut, int frameIndex) {   // :47-180
    const vec3 YCOEFF = {0.299f, 0.587f, 0.114f};
    float lCurr = dot(curr, YCOEFF);
    float lCurr2 = lCurr * lCurr;
    if (S.u.enableTAA == 0) return {curr.x, curr.y, curr.z, lCurr2};
    if (frameIndex == 0) return {curr.x, curr.y, curr.z, lCurr2};
    float motMag = length(motionOut);
    float MIN_W = S.u.taaHistoryMinWeight, AVG_W = S.u.taaHistoryAvgWeight, MAX_W = S.u.taaHistoryMaxWeight;
    float BOX = S.u.taaHistoryBoxSize;
    if (motMag < S.u.taaStillThresh) {
        vec4 pr = fetchPrev(S, uvCurr.x, uvCurr.y);
        vec3 prevCol = {pr.x, pr.y, pr.z};
        float wHist = (frameIndex < 8) ? MIN_W : ((frameIndex < 32) ? AVG_W : MAX_W);
        float wCurr = 1.0f - wHist;
        vec3 meanNew = prevCol * wHist + curr * wCurr;
        float m2New = pr.w * wHist + lCurr2 * wCurr;
        return {meanNew.x, meanNew.y, meanNew.z, m2New};
    }
    vec2 uvPrev = {uvCurr.x - motionOut.x * 0.5f, uvCurr.y - motionOut.y * 0.5f};
    bool oob = (uvPrev.x < 0.0f || uvPrev.y < 0.0f) || (uvPrev.x > 1.0f || uvPrev.y > 1.0f);
    if (oob) return {curr.x, curr.y, curr.z, lCurr2};
    vec4 pr = fetchPrev(S, uvPrev.x, uvPrev.y);
    vec3 prevCol = {pr.x, pr.y, pr.z};
    float wHist = 1.0f - smoothstepf(0.02f, S.u.taaHardMovingThresh, motMag);
    if (motMag > S.u.taaHardMovingThresh) wHist = 0.0f;
    float lPrev = dot(prevCol, YCOEFF);
    float maxL = fmax_(fmax_(lCurr, lPrev), 1e-3f);
    float relDiff = std::fabs(lCurr - lPrev) / maxL;
    float colorWeight = 1.0f - smoothstepf(0.03f, 0.25f, relDiff);
    wHist *= colorWeight;
    bool bigColorChange = (motMag > 0.02f) && (relDiff > 0.30f);
    if (bigColorChange) wHist = 0.0f;
    wHist = clampf(wHist, 0.0f, MAX_W);
    float wCurr = 1.0f - wHist;
    vec3 lo = curr - v3(BOX), hi = curr + v3(BOX);
    vec3 hc = {clampf(prevCol.x, lo.x, hi.x), clampf(prevCol.y, lo.y, hi.y), clampf(prevCol.z, lo.z, hi.z)};
    vec3 taaCol = wHist * hc + wCurr * curr;
    float m2New = wHist * pr.w + wCurr * lCurr2;
    return {taaCol.x, taaCol.y, taaCol.z, m2New};
}

// ---------------------------------------------------------------- rt.frag main()
struct PixelOut { vec4 color; vec2 motion; vec4 gpos; vec4 gnrm; };

static PixelOut shadePixel(const Scene &S, Counters &C, int px, int py) {     // rt.frag:50-197
    const OrcUniforms &u = S.u;
    Frag F{{(float)px + 0.5f, (float)py + 0.5f}};
    int SPP = std::max(u.spp, 1);
    vec2 camJit = (u.enableJitter == 1) ? vec2{u.jitter[0], u.jitter[1]} : vec2{0.0f, 0.0f};
    vec2 uv = {(F.fc.x + camJit.x) / u.resolution[0], (F.fc.y + camJit.y) / u.resolution[1]};
    vec2 ndc = {uv.x * 2.0f - 1.0f, uv.y * 2.0f - 1.0f};
    vec3 camPos = ld3(u.camPos), camRight = ld3(u.camRight), camUp = ld3(u.camUp), camFwd = ld3(u.camFwd);
    vec3 dir = normalize(camFwd + (ndc.x * camRight) * (u.tanHalfFov * u.aspect) + (ndc.y * camUp) * u.tanHalfFov);

    vec3 frameSum = v3(0.0f);
    vec2 motionOut = {0.0f, 0.0f};
    PixelOut o;
    o.gpos = {0, 0, 0, 0};
    o.gnrm = {0, 0, 0, 0};
    for (int s = 0; s < SPP; ++s) {
        int seed = (int)((uint32_t)u.frameIndex * (uint32_t)SPP + (uint32_t)s);
        Hit h;
        uint64_t f0 = C.fetches();
        bool hitAny = (u.useBVH == 1) ? traceBVH(S, C, camPos, dir, h) : traceAnalytic(S, C, camPos, dir, h);
        C.fetchPrimary += C.fetches() - f0;
        vec3 radiance;
        if (hitAny) {
            if (s == 0) {
                C.hitPixels++;
                vec2 prevNDC = ndcFromWorld(h.p, u.prevViewProj);
                vec2 currNDC = ndcFromWorld(h.p, u.currViewProj);
                motionOut = {currNDC.x - prevNDC.x, currNDC.y - prevNDC.y};
                o.gpos = {h.p.x, h.p.y, h.p.z, 1.0f};
                vec3 nn = normalize(h.n);
                o.gnrm = {nn.x, nn.y, nn.z, 0.0f};
            }
            vec3 V = -dir;
            if (u.useBVH == 1) {
                radiance = directLightBVH(S, C, F, h, seed, V);
                if (u.enableGI == 1) radiance += u.giScaleBVH * oneBounceGIBVH(S, C, F, h, u.frameIndex, seed);
                if (u.enableAO == 1) radiance *= computeAO(S, C, F, h, u.frameIndex);
            } else {
                MaterialProps mat = getMaterial(S, h.mat);
                if (mat.type == 2) radiance = shadeGlass(S, C, F, h, V, mat, seed);
                else if (mat.type == 1) radiance = shadeMirror(S, C, F, h, V, mat, seed);
                else if (h.mat == MAT_POINTLIGHT_SPHERE) {
                    vec3 baseCol = ld3(u.pointLightColor) * u.pointLightIntensity;
                    float d = length(h.p - camPos);
                    float falloff = 1.0f / fmax_(d * d * 0.25f + 1.0f, 1.0f);
                    radiance = baseCol * falloff;
                } else {
                    radiance = directLight(S, C, F, h, seed, V);
                    if (u.enableGI == 1) radiance += u.giScaleAnalytic * oneBounceGIAnalytic(S, C, F, h, u.frameIndex, seed);
                    if (u.enableAO == 1) radiance *= computeAO(S, C, F, h, u.frameIndex);
                }
            }
        } else {
            radiance = sky(S, C, dir);
            if (u.cameraMoved == 1 && s == 0) motionOut = {4.0f, 4.0f};
        }
        frameSum += radiance;
    }
    vec3 curr = frameSum / (float)SPP;
    vec2 uvCurr = {((float)px + 0.5f) / (float)S.W, ((float)py + 0.5f) / (float)S.H};   // rt_fullscreen.vert:44
    vec2 taaMotion = (u.cameraMoved == 1) ? motionOut : vec2{0.0f, 0.0f};
    o.color = resolveTAA(S, curr, uvCurr, taaMotion, u.frameIndex);
    o.motion = motionOut;
    return o;
}

}  // namespace orc

using namespace orc;

extern "C" {

uint32_t orc_hash2(uint32_t x, uint32_t y) { return hash2(x, y); }
uint32_t orc_rand_bits(float px, float py, int frame) { return rand_bits(vec2{px, py}, frame); }
float orc_rand(float px, float py, int frame) { return rand_(vec2{px, py}, frame); }
void orc_ld2(int i, float *out) { vec2 v = ld2(i); out[0] = v.x; out[1] = v.y; }
float orc_sin(float x) { return sinf_(x); }
float orc_cos(float x) { return cosf_(x); }
float orc_pow(float x, float y) { return powf_(x, y); }
float orc_exp2(float x) { return exp2f_(x); }
float orc_log2(float x) { return log2f_(x); }
uint16_t orc_f32_to_f16(float f) { return f32_to_f16(f); }
float orc_f16_to_f32(uint16_t h) { return f16_to_f32(h); }
void orc_concentric(float pi, float ux, float uy, float *out) {
    Scene S; S.u.pi = pi; vec2 r = concentricSample(S, vec2{ux, uy}); out[0] = r.x; out[1] = r.y;
}
void orc_sample_hemisphere(float pi, const float *N, float ux, float uy, float *out) {
    Scene S; S.u.pi = pi; vec3 r = sampleHemisphereCosine(S, ld3(N), vec2{ux, uy}); out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
void orc_texture_cube(const uint8_t *faces, int faceSize, int ch, const float *dir, float *out) {
    Scene S; Counters C; S.env = faces; S.envSize = faceSize; S.envCh = ch; S.envFilter = g_envFilter;
    vec3 r = textureCube(S, C, ld3(dir)); out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
// Single closest-hit / any-hit queries against a reference-layout BVH (unit tests).
int orc_trace_bvh(const OrcUniforms *u, const float *nodes, const float *tris, const float *ro, const float *rd,
                  float *tOut, float *pOut, float *nOut, OrcCounters *c) {
    Scene S; S.u = *u; S.nodes = nodes; S.tris = tris;
    Counters C; Hit h; h.p = v3(0.0f);
    bool hit = traceBVH(S, C, ld3(ro), ld3(rd), h);
    if (hit) { *tOut = h.t; pOut[0] = h.p.x; pOut[1] = h.p.y; pOut[2] = h.p.z; nOut[0] = h.n.x; nOut[1] = h.n.y; nOut[2] = h.n.z; }
    if (c) { c->nodeFetch = C.nodeFetch; c->triFetch = C.triFetch; c->raysClosest = C.raysClosest; }
    return hit ? 1 : 0;
}
// aabbHit (rt_bvh.glsl:124-134) and triHit (:154-170) on their own, for the per-function vectors in tests/golden/glsl_bvh_kat.npz.
// out4 = {hit, tmin, tmax, 0};  out8 = {hit, t, n.x, n.y, n.z} (t, n only written on a hit, as in the shader).
void orc_aabb_hit(const float *ro, const float *rd, const float *bmin, const float *bmax, float *out4) {
    vec3 d = ld3(rd);
    vec3 rdInv = {1.0f / d.x, 1.0f / d.y, 1.0f / d.z};
    float a = 0.0f, b = 0.0f;
    bool h = aabbHit(ld3(ro), rdInv, ld3(bmin), ld3(bmax), a, b);
    out4[0] = h ? 1.0f : 0.0f; out4[1] = a; out4[2] = b; out4[3] = 0.0f;
}
void orc_tri_hit(const OrcUniforms *u, const float *ro, const float *rd, const float *tri12, float tMax, float *out5) {
    Scene S; S.u = *u;
    TriSOA T = {{tri12[0], tri12[1], tri12[2]}, {tri12[4], tri12[5], tri12[6]}, {tri12[8], tri12[9], tri12[10]}};
    float t = 0.0f; vec3 n = v3(0.0f);
    bool h = triHit(S, ld3(ro), ld3(rd), T, tMax, t, n);
    out5[0] = h ? 1.0f : 0.0f; out5[1] = t; out5[2] = n.x; out5[3] = n.y; out5[4] = n.z;
}
// The BVH branch of rt.frag:92-106 for a given hit (no traversal: callers pass nodeCount = 0, so every shadow / bounce /
// AO ray leaves unoccluded): radiance = directLightBVH + giScaleBVH * oneBounceGIBVH, times computeAO.  Per-function
// vectors for tests/golden/glsl_bvh_shade_kat.npz.  hits: n x 12 floats {p.xyz, fragX, n.xyz, fragY, V.xyz, seed}.
void orc_shade_bvh_hits(const OrcUniforms *u, const uint8_t *envFaces, int envFaceSize, int envChannels, const float *hits, int n, float *out3) {
    Scene S; S.u = *u; S.env = envFaces; S.envSize = envFaceSize; S.envCh = envChannels;
    Counters C;
    for (int i = 0; i < n; ++i) {
        const float *q = hits + (size_t)i * 12;
        Hit h; h.p = {q[0], q[1], q[2]}; h.n = {q[4], q[5], q[6]}; h.mat = 1; h.t = 1.0f;
        Frag F; F.fc = {q[3], q[7]};
        vec3 V = {q[8], q[9], q[10]};
        int seed = (int)q[11];
        vec3 radiance = directLightBVH(S, C, F, h, seed, V);
        if (u->enableGI == 1) radiance += u->giScaleBVH * oneBounceGIBVH(S, C, F, h, u->frameIndex, seed);
        if (u->enableAO == 1) radiance *= computeAO(S, C, F, h, u->frameIndex);
        out3[i * 3 + 0] = radiance.x; out3[i * 3 + 1] = radiance.y; out3[i * 3 + 2] = radiance.z;
    }
}
int orc_trace_bvh_shadow(const OrcUniforms *u, const float *nodes, const float *tris, const float *ro, const float *rd, float tMax) {
    Scene S; S.u = *u; S.nodes = nodes; S.tris = tris;
    Counters C;
    return traceBVHShadow(S, C, ld3(ro), ld3(rd), tMax) ? 1 : 0;
}

// Render the pixel rectangle [x0,x1) x [y0,y1) of one frame. Targets are full W x H arrays, row 0 =
// bottom row (glReadPixels order), RGBA16F / RG16F / RGBA16F / RGBA16F as half bit patterns.
// pixelMask (optional, W*H bytes): only pixels with a non-zero byte are shaded (tile-parallel tests).
// EXTENSION knob (see Scene::giBounces): bounces of the analytic / hybrid GI path for the orc_render calls that follow.  1 = the reference.
static std::atomic<int> g_giBounces{1};
void orc_set_gi_bounces(int n) { g_giBounces = n; }
void orc_set_env_filter(int m) { g_envFilter = m; }

int orc_render(const OrcUniforms *u, const float *nodes12, const float *tris12, const uint8_t *envFaces, int envFaceSize,
               int envChannels, const uint16_t *prevAccum, uint16_t *outColor, uint16_t *outMotion, uint16_t *outGPos,
               uint16_t *outGNrm, int x0, int y0, int x1, int y1, const uint8_t *pixelMask, int nthreads, OrcCounters *counters) {
    Scene S;
    S.u = *u;
    S.nodes = nodes12; S.tris = tris12;
    S.env = envFaces; S.envSize = envFaceSize; S.envCh = envChannels;
    S.prev = prevAccum;
    S.giBounces = g_giBounces;
    S.envFilter = g_envFilter;
    S.W = (int)u->resolution[0]; S.H = (int)u->resolution[1];
    if (S.W <= 0 || S.H <= 0) return -1;
    if (u->useEnvMap == 1 && (!envFaces || envFaceSize <= 0 || envChannels < 3)) return -2;
    if ((u->useBVH == 1 || u->useBVH == kSceneHybrid) && u->nodeCount > 0 && u->triCount > 0 && (!nodes12 || !tris12)) return -3;
    x0 = std::max(x0, 0); y0 = std::max(y0, 0); x1 = std::min(x1, S.W); y1 = std::min(y1, S.H);
    if (nthreads < 1) nthreads = 1;
    std::atomic<int> nextRow{y0};
    std::vector<Counters> perThread((size_t)nthreads);
    auto worker = [&](int tid) {
        Counters &C = perThread[(size_t)tid];
        for (;;) {
            int y = nextRow.fetch_add(1);
            if (y >= y1) break;
            for (int x = x0; x < x1; ++x) {
                size_t idx = (size_t)y * S.W + x;
                if (pixelMask && !pixelMask[idx]) continue;
                PixelOut o = shadePixel(S, C, x, y);
                if (outColor) { uint16_t *p = outColor + idx * 4; p[0] = f32_to_f16(o.color.x); p[1] = f32_to_f16(o.color.y); p[2] = f32_to_f16(o.color.z); p[3] = f32_to_f16(o.color.w); }
                if (outMotion) { uint16_t *p = outMotion + idx * 2; p[0] = f32_to_f16(o.motion.x); p[1] = f32_to_f16(o.motion.y); }
                if (outGPos) { uint16_t *p = outGPos + idx * 4; p[0] = f32_to_f16(o.gpos.x); p[1] = f32_to_f16(o.gpos.y); p[2] = f32_to_f16(o.gpos.z); p[3] = f32_to_f16(o.gpos.w); }
                if (outGNrm) { uint16_t *p = outGNrm + idx * 4; p[0] = f32_to_f16(o.gnrm.x); p[1] = f32_to_f16(o.gnrm.y); p[2] = f32_to_f16(o.gnrm.z); p[3] = f32_to_f16(o.gnrm.w); }
            }
        }
    };
    if (nthreads == 1) worker(0);
    else {
        std::vector<std::thread> th;
        for (int i = 0; i < nthreads; ++i) th.emplace_back(worker, i);
        for (auto &t : th) t.join();
    }
    if (counters) {
        OrcCounters c{};
        for (auto &C : perThread) {
            c.raysClosest += C.raysClosest; c.raysShadow += C.raysShadow; c.raysAnalytic += C.raysAnalytic;
            c.nodeFetch += C.nodeFetch; c.triFetch += C.triFetch; c.envLookup += C.envLookup; c.hitPixels += C.hitPixels;
            c.fetchPrimary += C.fetchPrimary; c.fetchShadow += C.fetchShadow; c.fetchAO += C.fetchAO;
        }
        *counters = c;
    }
    return 0;
}

}  // extern "C"

// ================================================================================================
// Present pass: shaders/rt/rt_present.frag (SVGF-lite 7x7 + ACES + gamma), uniforms as set at
// src/render/render.cpp:206-235.  Inputs are the four targets of the frame just rendered (NEAREST,
// CLAMP_TO_EDGE: src/render/accum.cpp:11-14, gbuffer.cpp:16-19); output = default framebuffer RGBA8.
namespace orc {

struct PresentIn {
    const uint16_t *color, *motion, *gpos, *gnrm;
    int W, H;
    OrcPresentParams p;
};
static vec4 tex4(const uint16_t *img, int W, int H, float u, float v) {   // texture(sampler2D, uv), NEAREST
    int x = (int)std::floor(u * (float)W), y = (int)std::floor(v * (float)H);
    x = std::min(std::max(x, 0), W - 1);
    y = std::min(std::max(y, 0), H - 1);
    const uint16_t *p = img + ((size_t)y * W + x) * 4;
    return {f16_to_f32(p[0]), f16_to_f32(p[1]), f16_to_f32(p[2]), f16_to_f32(p[3])};
}
static vec2 tex2(const uint16_t *img, int W, int H, float u, float v) {
    int x = (int)std::floor(u * (float)W), y = (int)std::floor(v * (float)H);
    x = std::min(std::max(x, 0), W - 1);
    y = std::min(std::max(y, 0), H - 1);
    const uint16_t *p = img + ((size_t)y * W + x) * 2;
    return {f16_to_f32(p[0]), f16_to_f32(p[1])};
}
static vec3 acesTonemap(vec3 x, float exposure) {                        // rt_present.frag:65-69
    x = x * exposure;
    const float a = 2.51f, b = 0.03f, c = 2.43f, d = 0.59f, e = 0.14f;
    vec3 num = x * (a * x + v3(b));
    vec3 den = x * (c * x + v3(d)) + v3(e);
    return {clampf(num.x / den.x, 0.0f, 1.0f), clampf(num.y / den.y, 0.0f, 1.0f), clampf(num.z / den.z, 0.0f, 1.0f)};
}
static vec3 hsv2rgb(vec3 c) {                                            // :74-77
    vec3 q = {fractf(c.x + 0.0f), fractf(c.x + 2.0f / 3.0f), fractf(c.x + 1.0f / 3.0f)};
    vec3 p = {std::fabs(q.x * 6.0f - 3.0f), std::fabs(q.y * 6.0f - 3.0f), std::fabs(q.z * 6.0f - 3.0f)};
    vec3 k = {clampf(p.x - 1.0f, 0.0f, 1.0f), clampf(p.y - 1.0f, 0.0f, 1.0f), clampf(p.z - 1.0f, 0.0f, 1.0f)};
    return c.z * mix(v3(1.0f), k, c.y);
}
static vec3 visualizeMotion(vec2 motion, float scale) {                  // :92-104
    vec2 m = {motion.x * scale, motion.y * scale};
    float mag = length(m);
    if (mag < 1e-4f) return v3(0.0f);
    float hue = atan2f_(m.y, m.x) / (2.0f * 3.1415926535f) + 0.5f;
    float val = clampf(mag, 0.0f, 1.0f);
    return hsv2rgb(v3(hue, 1.0f, val));
}
static vec3 svgfFilter(const PresentIn &I, float u, float v) {           // :126-225
    const OrcPresentParams &P = I.p;
    const vec3 Y = {0.299f, 0.587f, 0.114f};
    vec4 centerRaw = tex4(I.color, I.W, I.H, u, v);
    vec3 cCenter = {centerRaw.x, centerRaw.y, centerRaw.z};
    float lCenter = dot(cCenter, Y);
    float varCenter = fmax_(centerRaw.w - lCenter * lCenter, 0.0f);
    varCenter = fmin_(varCenter, P.varMax);
    vec2 motion = tex2(I.motion, I.W, I.H, u, v);
    float motMag = length(motion);
    vec4 pc = tex4(I.gpos, I.W, I.H, u, v), nc = tex4(I.gnrm, I.W, I.H, u, v);
    vec3 pCenter = {pc.x, pc.y, pc.z}, nCenter = {nc.x, nc.y, nc.z};
    float texelX = 1.0f / P.resolution[0], texelY = 1.0f / P.resolution[1];
    vec3 accumCol = v3(0.0f);
    float accumW = 0.0f;
    float t = clampf(smoothstepf(0.005f, 0.05f, motMag), 0.0f, 1.0f);
    float kVar = mixf(P.kVar, P.kVarMotion, t);
    float kColor = mixf(P.kColor, P.kColorMotion, t);
    const float K_NRM = 2.0f, K_POS = 0.02f;
    float varBoost = 1.0f + varCenter * (1.0f + kVar * 0.5f);
    for (int j = -3; j <= 3; ++j)
        for (int i = -3; i <= 3; ++i) {
            float un = u + (float)i * texelX, vn = v + (float)j * texelY;
            if (un < 0.0f || un > 1.0f || vn < 0.0f || vn > 1.0f) continue;
            vec4 s = tex4(I.color, I.W, I.H, un, vn);
            vec3 c = {s.x, s.y, s.z};
            vec3 dc = c - cCenter;
            float dc2 = dot(dc, dc);
            float wCol = expf_(-dc2 * (kColor * 0.3f + 0.05f));
            vec4 p4 = tex4(I.gpos, I.W, I.H, un, vn), n4 = tex4(I.gnrm, I.W, I.H, un, vn);
            vec3 p = {p4.x, p4.y, p4.z}, n = {n4.x, n4.y, n4.z};
            vec3 dp = p - pCenter;
            float dist2 = dot(dp, dp);
            float wPos = expf_(-dist2 * K_POS);
            float ndot = clampf(dot(normalize(nCenter), normalize(n)), -1.0f, 1.0f);
            float nDiff = fmax_(0.0f, 1.0f - ndot);
            float wNrm = expf_(-nDiff * K_NRM);
            float wSpatial = (i == 0 && j == 0) ? 1.0f : 1.0f + varCenter * 4.0f;
            float w = varBoost * wCol * wPos * wNrm * wSpatial;
            accumCol += c * w;
            accumW += w;
        }
    if (accumW <= 0.0f) return cCenter;
    return accumCol / accumW;
}
static uint8_t unorm8(float x) {   // GL float -> UNORM8 conversion: clamp, scale, round to nearest
    float c = clampf(x, 0.0f, 1.0f);
    if (c != c) c = 0.0f;
    return (uint8_t)std::rint(c * 255.0f);
}
static void presentPixel(const PresentIn &I, int px, int py, uint8_t *out) {   // main(), :231-266
    const OrcPresentParams &P = I.p;
    float u = (((float)px + 0.5f) + 0.5f) / (float)I.W, v = (((float)py + 0.5f) + 0.5f) / (float)I.H;   // (gl_FragCoord.xy + 0.5) / size
    vec3 rgb;
    if (P.showMotion == 1) {
        vec2 m = tex2(I.motion, I.W, I.H, u, v);
        rgb = visualizeMotion(m, P.motionScale);
    } else {
        vec4 rawc = tex4(I.color, I.W, I.H, u, v);
        vec3 raw = {rawc.x, rawc.y, rawc.z};
        vec3 linearColor;
        if (P.enableSVGF == 0) linearColor = raw;
        else {
            vec3 filtered = svgfFilter(I, u, v);
            float s = clampf(P.svgfStrength, 0.0f, 1.0f);
            linearColor = mix(raw, filtered, s);
        }
        vec3 mapped = acesTonemap(linearColor, P.exposure);
        rgb = {powf_(mapped.x, 1.0f / 2.2f), powf_(mapped.y, 1.0f / 2.2f), powf_(mapped.z, 1.0f / 2.2f)};
    }
    out[0] = unorm8(rgb.x); out[1] = unorm8(rgb.y); out[2] = unorm8(rgb.z); out[3] = 255;
}

}  // namespace orc

extern "C" int orc_present(const OrcPresentParams *p, const uint16_t *color, const uint16_t *motion, const uint16_t *gpos,
                           const uint16_t *gnrm, uint8_t *outRGBA8, int nthreads) {
    orc::PresentIn I;
    I.color = color; I.motion = motion; I.gpos = gpos; I.gnrm = gnrm; I.p = *p;
    I.W = (int)p->resolution[0]; I.H = (int)p->resolution[1];
    if (I.W <= 0 || I.H <= 0) return -1;
    std::atomic<int> next{0};
    auto worker = [&]() {
        for (;;) {
            int y = next.fetch_add(1);
            if (y >= I.H) break;
            for (int x = 0; x < I.W; ++x) orc::presentPixel(I, x, y, outRGBA8 + ((size_t)y * I.W + x) * 4);
        }
    };
    std::vector<std::thread> th;
    for (int i = 1; i < std::max(nthreads, 1); ++i) th.emplace_back(worker);
    worker();
    for (auto &t : th) t.join();
    return 0;
}
extern "C" float orc_exp(float x) { return orc::expf_(x); }
extern "C" float orc_atan2(float y, float x) { return orc::atan2f_(y, x); }

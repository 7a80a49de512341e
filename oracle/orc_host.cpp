// oracle/orc_host.cpp -- TEST INFRASTRUCTURE ONLY (parity oracle). Never part of the product.
//
// CPU restatement of the reference's HOST side of the ray-trace path: median-split BVH builder and
// texture-buffer packing (src/scene/bvh.cpp), camera matrices (src/io/Camera.cpp + the glm closed
// forms it calls), per-frame jitter / cameraMoved (src/app/application.cpp:28-47, 387-405), the
// uniform marshalling of renderRay (src/render/render.cpp:8-167), RenderParams defaults
// (include/render/RenderParams.h) and the 4x3-cross cube-map slicing (src/render/cubemap.cpp).
// glm (1.0.0-85-g2d4c4b4d, an absent submodule) is restated from its published closed forms:
// lookAtRH, perspectiveRH_NO, translate, scale, radians.  parity unpinned.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "orc_types.h"

namespace {

struct V3 { float x, y, z; float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); } };
static inline V3 vmin(V3 a, V3 b) { return {std::min(a.x, b.x), std::min(a.y, b.y), std::min(a.z, b.z)}; }   // glm::min = (y<x)?y:x
static inline V3 vmax(V3 a, V3 b) { return {std::max(a.x, b.x), std::max(a.y, b.y), std::max(a.z, b.z)}; }
static inline V3 vadd(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
static inline V3 vsub(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
static inline V3 vscale(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
static inline float vdot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline V3 vcross(V3 a, V3 b) { return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y}; }
static inline V3 vnorm(V3 a) { float inv = 1.0f / std::sqrt(vdot(a, a)); return vscale(a, inv); }   // glm::normalize = v * inversesqrt(dot)

struct Tri { V3 v0, e1, e2; };                       // include/scene/bvh.h:20-24
struct Node { V3 bMin, bMax; int left, right, first, count; };   // include/scene/bvh.h:79-91
struct BuildRef { int triIndex; V3 c; };             // src/scene/bvh.cpp:30-33

static V3 tri_min(const Tri &t) { V3 v1 = vadd(t.v0, t.e1), v2 = vadd(t.v0, t.e2); return vmin(t.v0, vmin(v1, v2)); }   // bvh.cpp:10-14
static V3 tri_max(const Tri &t) { V3 v1 = vadd(t.v0, t.e1), v2 = vadd(t.v0, t.e2); return vmax(t.v0, vmax(v1, v2)); }   // bvh.cpp:16-20
static V3 tri_centroid(const Tri &t) {                                                                                    // bvh.cpp:22-26
    V3 v1 = vadd(t.v0, t.e1), v2 = vadd(t.v0, t.e2);
    return vscale(vadd(vadd(t.v0, v1), v2), 1.0f / 3.0f);
}

static int build_recursive(std::vector<Node> &nodes, const std::vector<Tri> &tris, std::vector<BuildRef> &refs, int begin,
                           int end, int leafMax) {   // bvh.cpp:41-91
    V3 bMin{1e30f, 1e30f, 1e30f}, bMax{-1e30f, -1e30f, -1e30f};
    for (int i = begin; i < end; ++i) {
        const Tri &T = tris[(size_t)refs[(size_t)i].triIndex];
        bMin = vmin(bMin, tri_min(T));
        bMax = vmax(bMax, tri_max(T));
    }
    const int count = end - begin;
    const int myIndex = (int)nodes.size();
    nodes.push_back(Node{});
    nodes[(size_t)myIndex].bMin = bMin;
    nodes[(size_t)myIndex].bMax = bMax;
    if (count <= leafMax) {
        nodes[(size_t)myIndex].left = -1;
        nodes[(size_t)myIndex].right = -1;
        nodes[(size_t)myIndex].first = begin;
        nodes[(size_t)myIndex].count = count;
        return myIndex;
    }
    const V3 e = vsub(bMax, bMin);
    int axis = (e.x > e.y) ? ((e.x > e.z) ? 0 : 2) : ((e.y > e.z) ? 1 : 2);
    const int mid = (begin + end) / 2;
    std::nth_element(refs.begin() + begin, refs.begin() + mid, refs.begin() + end,
                     [axis](const BuildRef &a, const BuildRef &b) { return a.c[axis] < b.c[axis]; });
    const int leftIdx = build_recursive(nodes, tris, refs, begin, mid, leafMax);
    const int rightIdx = build_recursive(nodes, tris, refs, mid, end, leafMax);
    nodes[(size_t)myIndex].left = leftIdx;
    nodes[(size_t)myIndex].right = rightIdx;
    nodes[(size_t)myIndex].first = -1;
    nodes[(size_t)myIndex].count = 0;
    return myIndex;
}

static std::vector<Node> build_bvh(std::vector<Tri> &tris) {   // bvh.cpp:94-137
    std::vector<Node> nodes;
    if (tris.empty()) return nodes;
    std::vector<BuildRef> refs(tris.size());
    for (size_t i = 0; i < tris.size(); ++i) {
        refs[i].triIndex = (int)i;
        refs[i].c = tri_centroid(tris[i]);
    }
    nodes.reserve(tris.size() * 2);
    build_recursive(nodes, tris, refs, 0, (int)refs.size(), 8);
    std::vector<Tri> remapped;
    remapped.reserve(tris.size());
    std::vector<int> stack;
    stack.push_back(0);
    while (!stack.empty()) {
        const int n = stack.back();
        stack.pop_back();
        const Node node = nodes[(size_t)n];
        if (node.count > 0) {
            for (int i = 0; i < node.count; ++i) remapped.push_back(tris[(size_t)refs[(size_t)(node.first + i)].triIndex]);
            nodes[(size_t)n].first = (int)remapped.size() - node.count;
        } else {
            stack.push_back(node.left);
            stack.push_back(node.right);
        }
    }
    tris = std::move(remapped);
    return nodes;
}

// glm column-major 4x4, m[c*4 + r].
struct M4 { float m[16]; };
static M4 m4_identity() { M4 r{}; r.m[0] = r.m[5] = r.m[10] = r.m[15] = 1.0f; return r; }
static M4 m4_mul(const M4 &a, const M4 &b) {   // glm operator*: column j of result = a * column j of b
    M4 r{};
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 4; ++i)
            r.m[j * 4 + i] = a.m[0 * 4 + i] * b.m[j * 4 + 0] + a.m[1 * 4 + i] * b.m[j * 4 + 1] + a.m[2 * 4 + i] * b.m[j * 4 + 2] +
                             a.m[3 * 4 + i] * b.m[j * 4 + 3];
    return r;
}
static inline float radians(float d) { return d * 0.01745329251994329576923690768489f; }   // glm::radians

static void camera_vectors(const OrcCamera &c, V3 &front, V3 &right, V3 &up) {   // src/io/Camera.cpp:54-63
    V3 f;
    f.x = std::cos(radians(c.yaw)) * std::cos(radians(c.pitch));
    f.y = std::sin(radians(c.pitch));
    f.z = std::sin(radians(c.yaw)) * std::cos(radians(c.pitch));
    front = vnorm(f);
    right = vnorm(vcross(front, V3{0.0f, 1.0f, 0.0f}));
    up = vnorm(vcross(right, front));
}
static M4 look_at(V3 eye, V3 center, V3 upv) {   // glm::lookAtRH
    V3 f = vnorm(vsub(center, eye));
    V3 s = vnorm(vcross(f, upv));
    V3 u = vcross(s, f);
    M4 r = m4_identity();
    r.m[0] = s.x; r.m[4] = s.y; r.m[8] = s.z;
    r.m[1] = u.x; r.m[5] = u.y; r.m[9] = u.z;
    r.m[2] = -f.x; r.m[6] = -f.y; r.m[10] = -f.z;
    r.m[12] = -vdot(s, eye); r.m[13] = -vdot(u, eye); r.m[14] = vdot(f, eye);
    return r;
}
static M4 perspective(float fovy, float aspect, float zn, float zf) {   // glm::perspectiveRH_NO
    const float t = std::tan(fovy / 2.0f);
    M4 r{};
    r.m[0] = 1.0f / (aspect * t);
    r.m[5] = 1.0f / t;
    r.m[10] = -(zf + zn) / (zf - zn);
    r.m[11] = -1.0f;
    r.m[14] = -(2.0f * zf * zn) / (zf - zn);
    return r;
}

static float host_halton(int index, int base) {   // src/app/application.cpp:28-38 (f *= 0.5f for every base: reproduced)
    float f = 1.0f, r = 0.0f;
    while (index > 0) {
        f *= 0.5f;
        const int digit = index % base;
        r += f * (float)digit;
        index /= base;
    }
    return r;
}

static V3 dirFromYawPitch(float yawDeg, float pitchDeg) {   // src/render/render.cpp:35-51
    float yaw = radians(yawDeg), pitch = radians(pitchDeg);
    float cp = std::cos(pitch), sp = std::sin(pitch), cy = std::cos(yaw), sy = std::sin(yaw);
    V3 d{cp * cy, sp, cp * sy};
    if (vdot(d, d) < 1e-6f) return {0.0f, -1.0f, 0.0f};
    return vnorm(d);
}
static V3 computePointLightWorldPos(const OrcRenderParams &p) {   // src/render/render.cpp:8-31
    V3 base{p.pointLightPos[0], p.pointLightPos[1], p.pointLightPos[2]};
    if (!p.pointLightOrbitEnabled || p.pointLightOrbitRadius <= 0.0f) return base;
    float yawRad = radians(p.pointLightYaw), pitchRad = radians(p.pointLightPitch);
    float cy = cosf(yawRad), sy = sinf(yawRad), cp = cosf(pitchRad), sp = sinf(pitchRad);
    V3 dir{cp * sy, sp, cp * cy};
    return vadd(base, vscale(dir, p.pointLightOrbitRadius));
}

}  // namespace

extern "C" {

// include/render/RenderParams.h:20-238
void orc_default_render_params(OrcRenderParams *p) {
    std::memset(p, 0, sizeof(*p));
    p->sppPerFrame = 1; p->exposure = 1.0f;
    p->matAlbedoColor[0] = 0.85f; p->matAlbedoColor[1] = 0.25f; p->matAlbedoColor[2] = 0.25f;
    p->matAlbedoSpecStrength = 0.35f; p->matAlbedoGloss = 48.0f;
    p->matGlassEnabled = 1; p->matGlassColor[0] = 0.95f; p->matGlassColor[1] = 0.98f; p->matGlassColor[2] = 1.0f;
    p->matGlassIOR = 1.5f; p->matGlassDistortion = 0.05f;
    p->matMirrorEnabled = 1; p->matMirrorColor[0] = p->matMirrorColor[1] = p->matMirrorColor[2] = 1.0f; p->matMirrorGloss = 256.0f;
    p->enableJitter = 1; p->jitterStillScale = 0.25f; p->jitterMovingScale = 0.5f;
    p->enableGI = 1; p->giScaleAnalytic = 0.35f; p->giScaleBVH = 0.20f;
    p->enableEnvMap = 1; p->envMapIntensity = 1.0f;
    p->sunEnabled = 1; p->sunColor[0] = 1.0f; p->sunColor[1] = 0.95f; p->sunColor[2] = 0.85f; p->sunIntensity = 0.45f;
    p->sunYaw = 45.0f; p->sunPitch = -35.0f;
    p->skyEnabled = 1; p->skyColor[0] = 0.4f; p->skyColor[1] = 0.5f; p->skyColor[2] = 1.0f; p->skyIntensity = 1.0f;
    p->skyYaw = 0.0f; p->skyPitch = 90.0f;
    p->pointLightEnabled = 1; p->pointLightColor[0] = 1.0f; p->pointLightColor[1] = 0.9f; p->pointLightColor[2] = 0.7f;
    p->pointLightIntensity = 20.0f; p->pointLightPos[0] = 0.0f; p->pointLightPos[1] = 2.5f; p->pointLightPos[2] = -3.0f;
    p->pointLightOrbitEnabled = 0; p->pointLightOrbitRadius = 3.5f; p->pointLightOrbitSpeed = 20.0f;
    p->pointLightYaw = 0.0f; p->pointLightPitch = 0.0f;
    p->enableAO = 1; p->aoSamples = 4; p->aoRadius = 0.8f; p->aoBias = 2e-3f; p->aoMin = 0.5f;
    p->enableTAA = 1; p->taaStillThresh = 1e-5f; p->taaHardMovingThresh = 0.35f; p->taaHistoryMinWeight = 0.85f;
    p->taaHistoryAvgWeight = 0.92f; p->taaHistoryMaxWeight = 0.96f; p->taaHistoryBoxSize = 0.06f;
    p->enableSVGF = 1; p->svgfVarMax = 0.05f; p->svgfKVar = 1.0f; p->svgfKColor = 1.2f; p->svgfKVarMotion = 0.8f;
    p->svgfKColorMotion = 1.5f; p->svgfStrength = 0.7f;
    p->motionScale = 4.0f;
}

// include/app/state.h:129-131
void orc_default_camera(OrcCamera *c) {
    c->pos[0] = 0.0f; c->pos[1] = 2.0f; c->pos[2] = 8.0f;
    c->yaw = -90.0f; c->pitch = -10.0f; c->fov = 60.0f; c->aspect = 1920.0f / 1080.0f;
}
// include/app/state.h:26-31: translate(-2,1.5,0) * scale(0.5), column-major.
void orc_default_bvh_transform(float *M) {
    M4 r = m4_identity();
    r.m[12] = -2.0f; r.m[13] = 1.5f; r.m[14] = 0.0f;     // glm::translate(I, v)
    r.m[0] *= 0.5f; r.m[5] *= 0.5f; r.m[10] *= 0.5f;     // glm::scale(M, 0.5): columns 0..2 scaled
    std::memcpy(M, r.m, sizeof(r.m));
}
void orc_camera_view(const OrcCamera *c, float *V) {   // Camera.cpp:66-68
    V3 f, r, u;
    camera_vectors(*c, f, r, u);
    V3 e{c->pos[0], c->pos[1], c->pos[2]};
    M4 m = look_at(e, vadd(e, f), u);
    std::memcpy(V, m.m, sizeof(m.m));
}
void orc_camera_proj(const OrcCamera *c, float *P) {   // Camera.cpp:71-73
    M4 m = perspective(radians(c->fov), c->aspect, 0.1f, 100.0f);
    std::memcpy(P, m.m, sizeof(m.m));
}
void orc_mat4_mul(const float *A, const float *B, float *out) {   // FrameState::beginFrame: P * V (frame_state.h:71)
    M4 a, b;
    std::memcpy(a.m, A, 64); std::memcpy(b.m, B, 64);
    M4 r = m4_mul(a, b);
    std::memcpy(out, r.m, 64);
}
// src/app/application.cpp:42-47
void orc_generate_jitter(int frameIndex, float *out) {
    const int idx = frameIndex & 1023;
    out[0] = host_halton(idx + 1, 2) - 0.5f;
    out[1] = host_halton(idx + 1, 3) - 0.5f;
}
// src/app/application.cpp:387-395
int orc_camera_moved(const float *currVP, const float *prevVP) {
    float vpDiff = 0.0f;
    for (int i = 0; i < 16; ++i) vpDiff = std::max(vpDiff, std::fabs(currVP[i] - prevVP[i]));
    return vpDiff > 1e-5f ? 1 : 0;
}

// renderRay's uniform block, src/render/render.cpp:67-167, plus the jitter policy of
// src/app/application.cpp:398-405.  envLoaded = (app.envMapTex != 0).
void orc_make_uniforms(const OrcRenderParams *p, const OrcCamera *cam, const float *currView, const float *currViewProj,
                       const float *prevViewProj, int fbw, int fbh, int frameIndex, int cameraMoved, int useBVH, int showMotion,
                       int nodeCount, int triCount, int envLoaded, OrcUniforms *u) {
    std::memset(u, 0, sizeof(*u));
    const float *V = currView;
    V3 right = vnorm(V3{V[0], V[4], V[8]});            // render.cpp:67  (currView[c][0])
    V3 up = vnorm(V3{V[1], V[5], V[9]});               // :68
    V3 fw = vnorm(V3{V[2], V[6], V[10]});              // :69
    V3 fwd{-fw.x, -fw.y, -fw.z};
    u->eps = 1e-4f; u->pi = 3.1415926535f; u->inf = 1e30f;   // RenderParams.h:229-231
    u->camPos[0] = cam->pos[0]; u->camPos[1] = cam->pos[1]; u->camPos[2] = cam->pos[2];
    u->camRight[0] = right.x; u->camRight[1] = right.y; u->camRight[2] = right.z;
    u->camUp[0] = up.x; u->camUp[1] = up.y; u->camUp[2] = up.z;
    u->camFwd[0] = fwd.x; u->camFwd[1] = fwd.y; u->camFwd[2] = fwd.z;
    u->tanHalfFov = tanf(radians(cam->fov) * 0.5f);    // :70
    u->aspect = cam->aspect;
    u->frameIndex = frameIndex;
    u->spp = showMotion ? 1 : p->sppPerFrame;          // :81
    u->resolution[0] = (float)fbw; u->resolution[1] = (float)fbh;
    if (p->enableJitter) {                             // application.cpp:398-405
        float j[2];
        orc_generate_jitter(frameIndex, j);
        const float scale = cameraMoved ? p->jitterMovingScale : p->jitterStillScale;
        u->jitter[0] = j[0] * scale; u->jitter[1] = j[1] * scale;
    }
    u->enableJitter = p->enableJitter ? 1 : 0;
    u->useBVH = useBVH == 2 ? 2 : (useBVH ? 1 : 0); u->nodeCount = nodeCount; u->triCount = triCount;   // render.cpp:94; 2 = the hybrid extension (rt_oracle.cpp kSceneHybrid)
    u->showMotion = showMotion ? 1 : 0;
    std::memcpy(u->prevViewProj, prevViewProj, 64);
    std::memcpy(u->currViewProj, currViewProj, 64);
    u->cameraMoved = cameraMoved ? 1 : 0;
    u->taaStillThresh = p->taaStillThresh; u->taaHardMovingThresh = p->taaHardMovingThresh;
    u->taaHistoryMinWeight = p->taaHistoryMinWeight; u->taaHistoryAvgWeight = p->taaHistoryAvgWeight;
    u->taaHistoryMaxWeight = p->taaHistoryMaxWeight; u->taaHistoryBoxSize = p->taaHistoryBoxSize;
    u->enableTAA = p->enableTAA;
    u->giScaleAnalytic = p->giScaleAnalytic; u->giScaleBVH = p->giScaleBVH;
    u->enableGI = p->enableGI; u->enableAO = p->enableAO; u->aoSamples = p->aoSamples;
    u->aoRadius = p->aoRadius; u->aoBias = p->aoBias; u->aoMin = p->aoMin;
    u->useEnvMap = (p->enableEnvMap && envLoaded) ? 1 : 0;   // :102
    u->envIntensity = p->envMapIntensity;
    V3 sunDir = dirFromYawPitch(p->sunYaw, p->sunPitch);    // :149-153
    u->sunEnabled = p->sunEnabled;
    std::memcpy(u->sunColor, p->sunColor, 12); u->sunIntensity = p->sunIntensity;
    u->sunDir[0] = sunDir.x; u->sunDir[1] = sunDir.y; u->sunDir[2] = sunDir.z;
    V3 skyDir = dirFromYawPitch(p->skyYaw, p->skyPitch);    // :156-160
    u->skyEnabled = p->skyEnabled;
    std::memcpy(u->skyColor, p->skyColor, 12); u->skyIntensity = p->skyIntensity;
    u->skyUpDir[0] = skyDir.x; u->skyUpDir[1] = skyDir.y; u->skyUpDir[2] = skyDir.z;
    V3 pp = computePointLightWorldPos(*p);                  // :163-167
    u->pointLightEnabled = p->pointLightEnabled;
    u->pointLightPos[0] = pp.x; u->pointLightPos[1] = pp.y; u->pointLightPos[2] = pp.z;
    std::memcpy(u->pointLightColor, p->pointLightColor, 12); u->pointLightIntensity = p->pointLightIntensity;
    std::memcpy(u->matAlbedoColor, p->matAlbedoColor, 12);  // :86-99
    u->matAlbedoSpecStrength = p->matAlbedoSpecStrength; u->matAlbedoGloss = p->matAlbedoGloss;
    std::memcpy(u->matGlassAlbedo, p->matGlassColor, 12);
    u->matGlassIOR = p->matGlassIOR; u->matGlassDistortion = p->matGlassDistortion; u->matGlassEnabled = p->matGlassEnabled;
    std::memcpy(u->matMirrorAlbedo, p->matMirrorColor, 12);
    u->matMirrorGloss = p->matMirrorGloss; u->matMirrorEnabled = p->matMirrorEnabled;
}

// gather_model_triangles (src/scene/bvh.cpp:225-246): positions[nVerts*3], indices[nIdx], model matrix M
// (column-major).  Writes 9 floats per triangle: v0, e1, e2.  Returns the triangle count.
int orc_gather_triangles(const float *positions, const uint32_t *indices, int nIdx, const float *M, float *outTris9) {
    int n = 0;
    auto xf = [&](uint32_t vi) {
        const float *p = positions + (size_t)vi * 3;
        // glm mat4 * vec4: ((col0*x + col1*y) + (col2*z + col3*w))
        V3 r;
        r.x = (M[0] * p[0] + M[4] * p[1]) + (M[8] * p[2] + M[12] * 1.0f);
        r.y = (M[1] * p[0] + M[5] * p[1]) + (M[9] * p[2] + M[13] * 1.0f);
        r.z = (M[2] * p[0] + M[6] * p[1]) + (M[10] * p[2] + M[14] * 1.0f);
        return r;
    };
    for (int k = 0; k + 2 < nIdx; k += 3) {
        V3 p0 = xf(indices[k]), p1 = xf(indices[k + 1]), p2 = xf(indices[k + 2]);
        V3 e1 = vsub(p1, p0), e2 = vsub(p2, p0);
        float *o = outTris9 + (size_t)n * 9;
        o[0] = p0.x; o[1] = p0.y; o[2] = p0.z; o[3] = e1.x; o[4] = e1.y; o[5] = e1.z; o[6] = e2.x; o[7] = e2.y; o[8] = e2.z;
        ++n;
    }
    return n;
}

// build_bvh + upload_bvh_tbo's packing (src/scene/bvh.cpp:94-221).  tris9 in: nTris*9 floats.
// Out: nodes12 (capacity 2*nTris*12 floats) and tris12 (nTris*12 floats, DFS-remapped order).
// Returns the node count.
int orc_build_bvh(const float *tris9, int nTris, float *nodes12, float *tris12) {
    std::vector<Tri> tris((size_t)nTris);
    for (int i = 0; i < nTris; ++i) {
        const float *p = tris9 + (size_t)i * 9;
        tris[(size_t)i] = Tri{{p[0], p[1], p[2]}, {p[3], p[4], p[5]}, {p[6], p[7], p[8]}};
    }
    std::vector<Node> nodes = build_bvh(tris);
    for (size_t i = 0; i < nodes.size(); ++i) {   // bvh.cpp:153-168
        const Node &n = nodes[i];
        float *o = nodes12 + i * 12;
        o[0] = n.bMin.x; o[1] = n.bMin.y; o[2] = n.bMin.z; o[3] = (float)n.left;
        o[4] = n.bMax.x; o[5] = n.bMax.y; o[6] = n.bMax.z; o[7] = (float)n.right;
        o[8] = (float)n.first; o[9] = (float)n.count; o[10] = 0.0f; o[11] = 0.0f;
    }
    for (size_t i = 0; i < tris.size(); ++i) {    // bvh.cpp:189-204
        const Tri &t = tris[i];
        float *o = tris12 + i * 12;
        o[0] = t.v0.x; o[1] = t.v0.y; o[2] = t.v0.z; o[3] = 0.0f;
        o[4] = t.e1.x; o[5] = t.e1.y; o[6] = t.e1.z; o[7] = 0.0f;
        o[8] = t.e2.x; o[9] = t.e2.y; o[10] = t.e2.z; o[11] = 0.0f;
    }
    return (int)nodes.size();
}

// loadCubeMapFromCross's slicing (src/render/cubemap.cpp:47-91): image rows top-to-bottom as decoded
// (no flip, :39); faces out in GL order +X -X +Y -Y +Z -Z, faceSize*faceSize*channels bytes each.
// Returns faceSize, or 0 for an invalid cross.
int orc_cubemap_from_cross(const uint8_t *img, int width, int height, int channels, uint8_t *faces) {
    if ((height % 3) != 0 || (width % 4) != 0 || (width / 4) != (height / 3)) return 0;
    const int faceSize = height / 3;
    const int stride = width * channels;
    const int ox[6] = {2, 0, 1, 1, 1, 3}, oy[6] = {1, 1, 0, 2, 1, 1};
    for (int f = 0; f < 6; ++f)
        for (int y = 0; y < faceSize; ++y)
            std::memcpy(faces + ((size_t)f * faceSize + y) * faceSize * channels,
                        img + (size_t)(oy[f] * faceSize + y) * stride + (size_t)ox[f] * faceSize * channels,
                        (size_t)faceSize * channels);
    return faceSize;
}

int orc_sizeof_uniforms(void) { return (int)sizeof(OrcUniforms); }
int orc_sizeof_render_params(void) { return (int)sizeof(OrcRenderParams); }

}  // extern "C"

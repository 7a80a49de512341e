// rt_host.cpp -- host half of librt_mi355.so (C++17, no GPU, no third-party math).
//
// Everything the reference does on the CPU for the ray-trace pass, re-stated for a headless
// library: scene build (src/scene/bvh.cpp), camera + frame state (src/io/Camera.cpp,
// include/render/frame_state.h, src/app/application.cpp:28-47/387-405), uniform marshalling
// (src/render/render.cpp:8-167), parameter defaults (include/render/RenderParams.h), asset
// readers standing in for Assimp / stb_image (.obj, .png) and the cube-map cross slicing
// (src/render/cubemap.cpp:47-91).  Compiled with -ffp-contract=off: the arrays produced here are
// compared bit for bit with the parity oracle's.
#include <zlib.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/rt_mi355.h"

namespace rthost {

struct F3 {
    float v[3];
    float &operator[](int i) { return v[i]; }
    float operator[](int i) const { return v[i]; }
};
static inline F3 f3(float x, float y, float z) { return F3{{x, y, z}}; }
static inline F3 add(const F3 &a, const F3 &b) { return f3(a[0] + b[0], a[1] + b[1], a[2] + b[2]); }
static inline F3 sub(const F3 &a, const F3 &b) { return f3(a[0] - b[0], a[1] - b[1], a[2] - b[2]); }
static inline F3 mul(const F3 &a, float s) { return f3(a[0] * s, a[1] * s, a[2] * s); }
static inline F3 lo(const F3 &a, const F3 &b) { return f3(std::min(a[0], b[0]), std::min(a[1], b[1]), std::min(a[2], b[2])); }
static inline F3 hi(const F3 &a, const F3 &b) { return f3(std::max(a[0], b[0]), std::max(a[1], b[1]), std::max(a[2], b[2])); }
static inline float dot3(const F3 &a, const F3 &b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static inline F3 cross3(const F3 &a, const F3 &b) {
    return f3(a[1] * b[2] - b[1] * a[2], a[2] * b[0] - b[2] * a[0], a[0] * b[1] - b[0] * a[1]);
}
static inline F3 unit(const F3 &a) { return mul(a, 1.0f / std::sqrt(dot3(a, a))); }
static inline float deg2rad(float d) { return d * 0.01745329251994329576923690768489f; }

// ------------------------------------------------------------------------------------------------
// Scene build.  CPU_Triangle{v0,e1,e2} (include/scene/bvh.h:20-24) is kept as 9 floats.
struct TriRef { int tri; F3 centroid; };

struct BuildNode { F3 bmin, bmax; int left = -1, right = -1, first = -1, count = 0; };

static void tri_bounds(const float *t, F3 &mn, F3 &mx, F3 &cen) {
    const F3 v0 = f3(t[0], t[1], t[2]);
    const F3 v1 = add(v0, f3(t[3], t[4], t[5]));
    const F3 v2 = add(v0, f3(t[6], t[7], t[8]));
    mn = lo(v0, lo(v1, v2));                       // bvh.cpp:10-14
    mx = hi(v0, hi(v1, v2));                       // bvh.cpp:16-20
    cen = mul(add(add(v0, v1), v2), 1.0f / 3.0f);  // bvh.cpp:22-26
}

// Median split, leaf <= 8, largest-extent axis, std::nth_element on the centroid, pre-order node
// numbering (bvh.cpp:41-91) -- written as an explicit work list instead of recursion so that
// million-triangle scenes do not depend on the thread's stack size.  Numbering stays pre-order
// because the right range is pushed before the left one.
static void build_nodes(const float *tris9, std::vector<TriRef> &refs, std::vector<BuildNode> &nodes) {
    struct Work { int begin, end, parent; bool isRight; };
    std::vector<Work> todo;
    todo.push_back({0, (int)refs.size(), -1, false});
    const int leafMax = 8;
    while (!todo.empty()) {
        const Work w = todo.back();
        todo.pop_back();
        F3 bmin = f3(1e30f, 1e30f, 1e30f), bmax = f3(-1e30f, -1e30f, -1e30f);
        for (int i = w.begin; i < w.end; ++i) {
            F3 mn, mx, c;
            tri_bounds(tris9 + (size_t)refs[(size_t)i].tri * 9, mn, mx, c);
            bmin = lo(bmin, mn);
            bmax = hi(bmax, mx);
        }
        const int self = (int)nodes.size();
        nodes.emplace_back();
        nodes[(size_t)self].bmin = bmin;
        nodes[(size_t)self].bmax = bmax;
        if (w.parent >= 0) (w.isRight ? nodes[(size_t)w.parent].right : nodes[(size_t)w.parent].left) = self;
        const int n = w.end - w.begin;
        if (n <= leafMax) {
            nodes[(size_t)self].first = w.begin;
            nodes[(size_t)self].count = n;
            continue;
        }
        const F3 e = sub(bmax, bmin);
        const int axis = (e[0] > e[1]) ? ((e[0] > e[2]) ? 0 : 2) : ((e[1] > e[2]) ? 1 : 2);   // bvh.cpp:72
        const int mid = (w.begin + w.end) / 2;
        std::nth_element(refs.begin() + w.begin, refs.begin() + mid, refs.begin() + w.end,
                         [axis](const TriRef &a, const TriRef &b) { return a.centroid[axis] < b.centroid[axis]; });
        todo.push_back({mid, w.end, self, true});
        todo.push_back({w.begin, mid, self, false});
    }
}

// ------------------------------------------------------------------------------------------------
// 4x4 column-major helpers (glm closed forms: lookAtRH, perspectiveRH_NO, translate, scale).
static void mat_identity(float *m) { std::memset(m, 0, 64); m[0] = m[5] = m[10] = m[15] = 1.0f; }
static void mat_mul(const float *a, const float *b, float *out) {
    float r[16];
    for (int col = 0; col < 4; ++col)
        for (int row = 0; row < 4; ++row)
            r[col * 4 + row] = a[0 * 4 + row] * b[col * 4 + 0] + a[1 * 4 + row] * b[col * 4 + 1] +
                               a[2 * 4 + row] * b[col * 4 + 2] + a[3 * 4 + row] * b[col * 4 + 3];
    std::memcpy(out, r, 64);
}
static void camera_axes(const RtCamera &c, F3 &front, F3 &right, F3 &up) {   // Camera::UpdateCameraVectors, Camera.cpp:54-63
    const float cy = std::cos(deg2rad(c.yaw)), sy = std::sin(deg2rad(c.yaw));
    const float cp = std::cos(deg2rad(c.pitch)), sp = std::sin(deg2rad(c.pitch));
    front = unit(f3(cy * cp, sp, sy * cp));
    right = unit(cross3(front, f3(0.0f, 1.0f, 0.0f)));
    up = unit(cross3(right, front));
}
static F3 dir_from_yaw_pitch(float yawDeg, float pitchDeg) {   // render.cpp:35-51
    const float yaw = deg2rad(yawDeg), pitch = deg2rad(pitchDeg);
    const float cp = std::cos(pitch), sp = std::sin(pitch), cy = std::cos(yaw), sy = std::sin(yaw);
    const F3 d = f3(cp * cy, sp, cp * sy);
    if (dot3(d, d) < 1e-6f) return f3(0.0f, -1.0f, 0.0f);
    return unit(d);
}
static float app_halton(int index, int base) {   // application.cpp:28-38; "f *= 0.5f" for every base is the reference's behaviour
    float f = 1.0f, r = 0.0f;
    while (index > 0) {
        f *= 0.5f;
        r += f * (float)(index % base);
        index /= base;
    }
    return r;
}

// ------------------------------------------------------------------------------------------------
// PNG reader (8-bit, non-interlaced; colour types 0/2/3/4/6) on zlib -- stands in for stbi_load.
static uint32_t be32(const uint8_t *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
static int paeth(int a, int b, int c) {
    const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
    return (pa <= pb && pa <= pc) ? a : ((pb <= pc) ? b : c);
}
static int decode_png(const std::vector<uint8_t> &file, std::vector<uint8_t> &pix, int &W, int &H, int &CH) {
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (file.size() < 33 || std::memcmp(file.data(), sig, 8) != 0) return RT_ERR_IO;
    size_t pos = 8;
    int depth = 0, ctype = -1, interlace = 0;
    std::vector<uint8_t> idat, palette;
    W = H = 0;
    while (pos + 12 <= file.size()) {
        const uint32_t len = be32(&file[pos]);
        const uint8_t *type = &file[pos + 4];
        if (pos + 12 + (size_t)len > file.size()) return RT_ERR_IO;
        const uint8_t *data = &file[pos + 8];
        if (!std::memcmp(type, "IHDR", 4)) {
            if (len < 13) return RT_ERR_IO;
            W = (int)be32(data); H = (int)be32(data + 4);
            depth = data[8]; ctype = data[9]; interlace = data[12];
        } else if (!std::memcmp(type, "PLTE", 4)) palette.assign(data, data + len);
        else if (!std::memcmp(type, "IDAT", 4)) idat.insert(idat.end(), data, data + len);
        else if (!std::memcmp(type, "IEND", 4)) break;
        pos += 12 + (size_t)len;
    }
    if (W <= 0 || H <= 0 || depth != 8 || interlace != 0) return RT_ERR_UNSUPPORTED;
    int srcCh;
    switch (ctype) {
        case 0: srcCh = 1; break;
        case 2: srcCh = 3; break;
        case 3: srcCh = 1; break;
        case 4: srcCh = 2; break;
        case 6: srcCh = 4; break;
        default: return RT_ERR_UNSUPPORTED;
    }
    const size_t stride = (size_t)W * srcCh;
    // a corrupt or hostile IHDR must not make us allocate gigabytes: deflate expands at most ~1032:1, so the declared image
    // cannot be larger than that multiple of the compressed data actually present
    if ((size_t)W > ((size_t)1 << 24) || (size_t)H > ((size_t)1 << 24) || (stride + 1) * (size_t)H > idat.size() * 1032 + 65536) return RT_ERR_IO;
    std::vector<uint8_t> raw((stride + 1) * (size_t)H);
    uLongf rawLen = (uLongf)raw.size();
    if (uncompress(raw.data(), &rawLen, idat.data(), (uLong)idat.size()) != Z_OK || rawLen != raw.size()) return RT_ERR_IO;
    std::vector<uint8_t> img(stride * (size_t)H);
    for (int y = 0; y < H; ++y) {
        const uint8_t *src = &raw[(stride + 1) * (size_t)y];
        const int filter = src[0];
        ++src;
        uint8_t *dst = &img[stride * (size_t)y];
        const uint8_t *up = y ? dst - stride : nullptr;
        for (size_t i = 0; i < stride; ++i) {
            const int a = (i >= (size_t)srcCh) ? dst[i - srcCh] : 0;
            const int b = up ? up[i] : 0;
            const int c = (up && i >= (size_t)srcCh) ? up[i - srcCh] : 0;
            int pred;
            switch (filter) {
                case 0: pred = 0; break;
                case 1: pred = a; break;
                case 2: pred = b; break;
                case 3: pred = (a + b) >> 1; break;
                case 4: pred = paeth(a, b, c); break;
                default: return RT_ERR_IO;
            }
            dst[i] = (uint8_t)(src[i] + pred);
        }
    }
    if (ctype == 3) {   // palette -> RGB
        CH = 3;
        pix.resize((size_t)W * H * 3);
        for (size_t i = 0; i < (size_t)W * H; ++i) {
            const size_t k = (size_t)img[i] * 3;
            if (k + 2 >= palette.size()) return RT_ERR_IO;
            pix[i * 3 + 0] = palette[k]; pix[i * 3 + 1] = palette[k + 1]; pix[i * 3 + 2] = palette[k + 2];
        }
    } else {
        CH = srcCh;
        pix.swap(img);
    }
    return RT_OK;
}

// No C++ exception may cross the C ABI (include/rt_mi355.h): allocation failures and anything else thrown by the standard
// library inside an entry point become a status code.
template <class F> static int guarded(F &&body) {
    try { return body(); }
    catch (const std::bad_alloc &) { return RT_ERR_IO; }
    catch (...) { return RT_ERR_INVALID; }
}

}  // namespace rthost

using namespace rthost;

extern "C" {

const char *rt_version(void) { return "rt_mi355 0.1 (gfx950)"; }
int rt_sizeof_uniforms(void) { return (int)sizeof(RtUniforms); }
int rt_sizeof_render_params(void) { return (int)sizeof(RtRenderParams); }
void rt_free(void *p) { std::free(p); }

void rt_default_render_params(RtRenderParams *p) {
    static const RtRenderParams d = {
        /*sppPerFrame*/ 1, /*exposure*/ 1.0f,
        /*matAlbedo*/ {0.85f, 0.25f, 0.25f}, 0.35f, 48.0f,
        /*glass*/ 1, {0.95f, 0.98f, 1.0f}, 1.5f, 0.05f,
        /*mirror*/ 1, {1.0f, 1.0f, 1.0f}, 256.0f,
        /*jitter*/ 1, 0.25f, 0.5f,
        /*GI*/ 1, 0.35f, 0.20f,
        /*env*/ 1, 1.0f,
        /*sun*/ 1, {1.0f, 0.95f, 0.85f}, 0.45f, 45.0f, -35.0f,
        /*sky*/ 1, {0.4f, 0.5f, 1.0f}, 1.0f, 0.0f, 90.0f,
        /*point*/ 1, {1.0f, 0.9f, 0.7f}, 20.0f, {0.0f, 2.5f, -3.0f},
        /*orbit*/ 0, 3.5f, 20.0f, 0.0f, 0.0f,
        /*AO*/ 1, 4, 0.8f, 2e-3f, 0.5f,
        /*TAA*/ 1, 1e-5f, 0.35f, 0.85f, 0.92f, 0.96f, 0.06f,
        /*SVGF*/ 1, 0.05f, 1.0f, 1.2f, 0.8f, 1.5f, 0.7f,
        /*motionScale*/ 4.0f};
    *p = d;
}

void rt_default_camera(RtCamera *c) {
    const RtCamera d = {{0.0f, 2.0f, 8.0f}, -90.0f, -10.0f, 60.0f, 1920.0f / 1080.0f};
    *c = d;
}

void rt_default_bvh_transform(float *M) {
    // translate(I, (-2, 1.5, 0)) then scale(., 0.5): T * S, column-major.
    mat_identity(M);
    M[0] = 0.5f; M[5] = 0.5f; M[10] = 0.5f;
    M[12] = -2.0f; M[13] = 1.5f; M[14] = 0.0f;
}

void rt_camera_view(const RtCamera *c, float *V) {
    F3 front, right, up;
    camera_axes(*c, front, right, up);
    const F3 eye = f3(c->pos[0], c->pos[1], c->pos[2]);
    // lookAtRH(eye, eye + front, up)
    const F3 f = unit(sub(add(eye, front), eye));
    const F3 s = unit(cross3(f, up));
    const F3 u = cross3(s, f);
    mat_identity(V);
    V[0] = s[0]; V[4] = s[1]; V[8] = s[2];
    V[1] = u[0]; V[5] = u[1]; V[9] = u[2];
    V[2] = -f[0]; V[6] = -f[1]; V[10] = -f[2];
    V[12] = -dot3(s, eye); V[13] = -dot3(u, eye); V[14] = dot3(f, eye);
}

void rt_camera_proj(const RtCamera *c, float *P) {
    const float zn = 0.1f, zf = 100.0f;
    const float t = std::tan(deg2rad(c->fov) / 2.0f);
    std::memset(P, 0, 64);
    P[0] = 1.0f / (c->aspect * t);
    P[5] = 1.0f / t;
    P[10] = -(zf + zn) / (zf - zn);
    P[11] = -1.0f;
    P[14] = -(2.0f * zf * zn) / (zf - zn);
}

void rt_mat4_mul(const float *A, const float *B, float *out) { mat_mul(A, B, out); }

void rt_generate_jitter(int frameIndex, float *out) {
    const int idx = frameIndex & 1023;
    out[0] = app_halton(idx + 1, 2) - 0.5f;
    out[1] = app_halton(idx + 1, 3) - 0.5f;
}

int rt_camera_moved(const float *currVP, const float *prevVP) {
    float worst = 0.0f;
    for (int i = 0; i < 16; ++i) worst = std::max(worst, std::fabs(currVP[i] - prevVP[i]));
    return worst > 1e-5f;
}

void rt_make_uniforms(const RtRenderParams *p, const RtCamera *cam, const float *V, const float *currVP, const float *prevVP,
                      int fbw, int fbh, int frameIndex, int cameraMoved, int useBVH, int showMotion, int nodeCount,
                      int triCount, int envLoaded, RtUniforms *u) {
    std::memset(u, 0, sizeof(*u));
    auto put3 = [](float *dst, const F3 &s) { dst[0] = s[0]; dst[1] = s[1]; dst[2] = s[2]; };
    // camera basis: rows 0/1/2 of the view matrix (render.cpp:67-69)
    put3(u->camRight, unit(f3(V[0], V[4], V[8])));
    put3(u->camUp, unit(f3(V[1], V[5], V[9])));
    put3(u->camFwd, mul(unit(f3(V[2], V[6], V[10])), -1.0f));
    std::memcpy(u->camPos, cam->pos, 12);
    u->eps = 1e-4f; u->pi = 3.1415926535f; u->inf = 1e30f;
    u->tanHalfFov = tanf(deg2rad(cam->fov) * 0.5f);
    u->aspect = cam->aspect;
    u->frameIndex = frameIndex;
    u->spp = showMotion ? 1 : p->sppPerFrame;
    u->resolution[0] = (float)fbw; u->resolution[1] = (float)fbh;
    u->enableJitter = p->enableJitter ? 1 : 0;
    if (p->enableJitter) {
        float j[2];
        rt_generate_jitter(frameIndex, j);
        const float k = cameraMoved ? p->jitterMovingScale : p->jitterStillScale;
        u->jitter[0] = j[0] * k; u->jitter[1] = j[1] * k;
    }
    u->useBVH = useBVH == RT_SCENE_HYBRID ? RT_SCENE_HYBRID : (useBVH ? 1 : 0); u->nodeCount = nodeCount; u->triCount = triCount;   // render.cpp:94 sets app.useBVH ? 1 : 0; 2 = the hybrid extension
    u->showMotion = showMotion ? 1 : 0;
    std::memcpy(u->prevViewProj, prevVP, 64);
    std::memcpy(u->currViewProj, currVP, 64);
    u->cameraMoved = cameraMoved ? 1 : 0;
    u->taaStillThresh = p->taaStillThresh; u->taaHardMovingThresh = p->taaHardMovingThresh;
    u->taaHistoryMinWeight = p->taaHistoryMinWeight; u->taaHistoryAvgWeight = p->taaHistoryAvgWeight;
    u->taaHistoryMaxWeight = p->taaHistoryMaxWeight; u->taaHistoryBoxSize = p->taaHistoryBoxSize;
    u->enableTAA = p->enableTAA;
    u->giScaleAnalytic = p->giScaleAnalytic; u->giScaleBVH = p->giScaleBVH;
    u->enableGI = p->enableGI; u->enableAO = p->enableAO; u->aoSamples = p->aoSamples;
    u->aoRadius = p->aoRadius; u->aoBias = p->aoBias; u->aoMin = p->aoMin;
    u->useEnvMap = (p->enableEnvMap && envLoaded) ? 1 : 0;
    u->envIntensity = p->envMapIntensity;
    u->sunEnabled = p->sunEnabled; std::memcpy(u->sunColor, p->sunColor, 12); u->sunIntensity = p->sunIntensity;
    put3(u->sunDir, dir_from_yaw_pitch(p->sunYaw, p->sunPitch));
    u->skyEnabled = p->skyEnabled; std::memcpy(u->skyColor, p->skyColor, 12); u->skyIntensity = p->skyIntensity;
    put3(u->skyUpDir, dir_from_yaw_pitch(p->skyYaw, p->skyPitch));
    // computePointLightWorldPos, render.cpp:8-31
    F3 lp = f3(p->pointLightPos[0], p->pointLightPos[1], p->pointLightPos[2]);
    if (p->pointLightOrbitEnabled && p->pointLightOrbitRadius > 0.0f) {
        const float yr = deg2rad(p->pointLightYaw), pr = deg2rad(p->pointLightPitch);
        const float cy = cosf(yr), sy = sinf(yr), cp = cosf(pr), sp = sinf(pr);
        lp = add(lp, mul(f3(cp * sy, sp, cp * cy), p->pointLightOrbitRadius));
    }
    u->pointLightEnabled = p->pointLightEnabled; put3(u->pointLightPos, lp);
    std::memcpy(u->pointLightColor, p->pointLightColor, 12); u->pointLightIntensity = p->pointLightIntensity;
    std::memcpy(u->matAlbedoColor, p->matAlbedoColor, 12);
    u->matAlbedoSpecStrength = p->matAlbedoSpecStrength; u->matAlbedoGloss = p->matAlbedoGloss;
    std::memcpy(u->matGlassAlbedo, p->matGlassColor, 12);
    u->matGlassIOR = p->matGlassIOR; u->matGlassDistortion = p->matGlassDistortion; u->matGlassEnabled = p->matGlassEnabled;
    std::memcpy(u->matMirrorAlbedo, p->matMirrorColor, 12);
    u->matMirrorGloss = p->matMirrorGloss; u->matMirrorEnabled = p->matMirrorEnabled;
}

void rt_make_present_params(const RtRenderParams *p, int showMotion, int fbw, int fbh, RtPresentParams *o) {   // render.cpp:209-235
    o->exposure = p->exposure; o->showMotion = showMotion ? 1 : 0; o->motionScale = p->motionScale;
    o->resolution[0] = (float)fbw; o->resolution[1] = (float)fbh;
    o->varMax = p->svgfVarMax; o->kVar = p->svgfKVar; o->kColor = p->svgfKColor; o->kVarMotion = p->svgfKVarMotion;
    o->kColorMotion = p->svgfKColorMotion; o->svgfStrength = p->svgfStrength; o->enableSVGF = p->enableSVGF ? 1 : 0;
}

int rt_gather_triangles(const float *positions, const uint32_t *indices, int nIdx, const float *M, float *out) {
    if (!positions || !indices || !M || !out || nIdx < 0) return RT_ERR_INVALID;
    auto world = [&](uint32_t vi) {
        const float *p = positions + (size_t)vi * 3;
        F3 r;   // glm mat4*vec4 evaluation order: (c0*x + c1*y) + (c2*z + c3*w)
        for (int k = 0; k < 3; ++k) r[k] = (M[k] * p[0] + M[4 + k] * p[1]) + (M[8 + k] * p[2] + M[12 + k] * 1.0f);
        return r;
    };
    int n = 0;
    for (int k = 0; k + 2 < nIdx; k += 3, ++n) {
        const F3 a = world(indices[k]), b = world(indices[k + 1]), c = world(indices[k + 2]);
        const F3 e1 = sub(b, a), e2 = sub(c, a);
        float *o = out + (size_t)n * 9;
        for (int j = 0; j < 3; ++j) { o[j] = a[j]; o[3 + j] = e1[j]; o[6 + j] = e2[j]; }
    }
    return n;
}

int rt_gather_triangles_checked(const float *positions, int nVerts, const uint32_t *indices, int nIdx, const float *M, float *out) {
    if (!positions || !indices || !M || !out || nIdx < 0 || nVerts < 0) return RT_ERR_INVALID;
    for (int k = 0; k < nIdx; ++k)
        if (indices[k] >= (uint32_t)nVerts) return RT_ERR_INVALID;
    return rt_gather_triangles(positions, indices, nIdx, M, out);
}

int rt_build_bvh(const float *tris9, int nTris, float *nodes12, float *tris12) {
    if (nTris < 0 || (nTris > 0 && (!tris9 || !nodes12 || !tris12))) return RT_ERR_INVALID;
    if (nTris == 0) return 0;
    return guarded([&]() -> int {
    std::vector<TriRef> refs((size_t)nTris);
    for (int i = 0; i < nTris; ++i) {
        F3 mn, mx;
        refs[(size_t)i].tri = i;
        tri_bounds(tris9 + (size_t)i * 9, mn, mx, refs[(size_t)i].centroid);
    }
    std::vector<BuildNode> nodes;
    nodes.reserve((size_t)nTris * 2);
    build_nodes(tris9, refs, nodes);
    // Leaf re-packing, bvh.cpp:109-135: LIFO walk that pushes left then right, i.e. right subtree first.
    std::vector<int> order;
    order.reserve((size_t)nTris);
    std::vector<int> walk{0};
    while (!walk.empty()) {
        const int n = walk.back();
        walk.pop_back();
        BuildNode &nd = nodes[(size_t)n];
        if (nd.count > 0) {
            const int base = (int)order.size();
            for (int i = 0; i < nd.count; ++i) order.push_back(refs[(size_t)(nd.first + i)].tri);
            nd.first = base;
        } else {
            walk.push_back(nd.left);
            walk.push_back(nd.right);
        }
    }
    for (size_t i = 0; i < nodes.size(); ++i) {   // upload_bvh_tbo node texels, bvh.cpp:153-168
        const BuildNode &nd = nodes[i];
        float *o = nodes12 + i * 12;
        o[0] = nd.bmin[0]; o[1] = nd.bmin[1]; o[2] = nd.bmin[2]; o[3] = (float)nd.left;
        o[4] = nd.bmax[0]; o[5] = nd.bmax[1]; o[6] = nd.bmax[2]; o[7] = (float)nd.right;
        o[8] = (float)nd.first; o[9] = (float)nd.count; o[10] = 0.0f; o[11] = 0.0f;
    }
    for (size_t i = 0; i < order.size(); ++i) {   // triangle texels, bvh.cpp:189-204
        const float *t = tris9 + (size_t)order[i] * 9;
        float *o = tris12 + i * 12;
        o[0] = t[0]; o[1] = t[1]; o[2] = t[2]; o[3] = 0.0f;
        o[4] = t[3]; o[5] = t[4]; o[6] = t[5]; o[7] = 0.0f;
        o[8] = t[6]; o[9] = t[7]; o[10] = t[8]; o[11] = 0.0f;
    }
    return (int)nodes.size();
    });
}

int rt_load_obj(const char *path, float **positions, int *nVerts, uint32_t **indices, int *nIdx) {
    if (!path || !positions || !nVerts || !indices || !nIdx) return RT_ERR_INVALID;
    FILE *f = std::fopen(path, "rb");
    if (!f) return RT_ERR_IO;
    char *line = nullptr;          // getline: a face record may be arbitrarily long (fgets with a fixed buffer would cut it into wrong faces)
    size_t cap = 0;
    const int rc = guarded([&]() -> int {
    std::vector<float> pos;
    std::vector<uint32_t> idx;
    std::vector<long> face;
    while (getline(&line, &cap, f) >= 0) {
        const char *s = line;
        while (*s == ' ' || *s == '\t') ++s;
        if (s[0] == 'v' && (s[1] == ' ' || s[1] == '\t')) {
            float x = 0, y = 0, z = 0;
            if (std::sscanf(s + 2, "%f %f %f", &x, &y, &z) == 3) { pos.push_back(x); pos.push_back(y); pos.push_back(z); }
        } else if (s[0] == 'f' && (s[1] == ' ' || s[1] == '\t')) {
            face.clear();
            const char *q = s + 2;
            while (*q) {
                while (*q == ' ' || *q == '\t') ++q;
                if (*q == '\0' || *q == '\n' || *q == '\r' || *q == '#') break;
                char *end = nullptr;
                long v = std::strtol(q, &end, 10);   // "v", "v/vt", "v//vn", "v/vt/vn": only v is used
                if (end == q) break;
                const long nv = (long)(pos.size() / 3);
                if (v < 0) v = nv + v + 1;           // relative index
                if (v < 1 || v > nv) return RT_ERR_IO;
                face.push_back(v - 1);
                q = end;
                while (*q && *q != ' ' && *q != '\t' && *q != '\n' && *q != '\r') ++q;
            }
            for (size_t k = 1; k + 1 < face.size(); ++k) {   // fan, what aiProcess_Triangulate yields for convex faces
                idx.push_back((uint32_t)face[0]); idx.push_back((uint32_t)face[k]); idx.push_back((uint32_t)face[k + 1]);
            }
        }
    }
    *nVerts = (int)(pos.size() / 3);
    *nIdx = (int)idx.size();
    *positions = (float *)std::malloc(std::max<size_t>(pos.size(), 1) * sizeof(float));
    *indices = (uint32_t *)std::malloc(std::max<size_t>(idx.size(), 1) * sizeof(uint32_t));
    if (!*positions || !*indices) { std::free(*positions); std::free(*indices); *positions = nullptr; *indices = nullptr; return RT_ERR_IO; }
    std::memcpy(*positions, pos.data(), pos.size() * sizeof(float));
    std::memcpy(*indices, idx.data(), idx.size() * sizeof(uint32_t));
    return RT_OK;
    });
    std::free(line);
    std::fclose(f);
    return rc;
}

int rt_load_png(const char *path, uint8_t **pixels, int *width, int *height, int *channels) {
    if (!path || !pixels || !width || !height || !channels) return RT_ERR_INVALID;
    FILE *f = std::fopen(path, "rb");
    if (!f) return RT_ERR_IO;
    const int rc = guarded([&]() -> int {
        std::vector<uint8_t> file;
        uint8_t buf[65536];
        size_t got;
        while ((got = std::fread(buf, 1, sizeof buf, f)) > 0) file.insert(file.end(), buf, buf + got);
        std::vector<uint8_t> pix;
        const int rcd = decode_png(file, pix, *width, *height, *channels);
        if (rcd != RT_OK) return rcd;
        *pixels = (uint8_t *)std::malloc(pix.size());
        if (!*pixels) return RT_ERR_IO;
        std::memcpy(*pixels, pix.data(), pix.size());
        return RT_OK;
    });
    std::fclose(f);
    return rc;
}

int rt_save_png(const char *path, const uint8_t *pixels, int width, int height, int channels, int flipY) {
    if (!path || !pixels || width <= 0 || height <= 0 || channels < 1 || channels > 4) return RT_ERR_INVALID;
    return guarded([&]() -> int {
    static const uint8_t ctype[5] = {0, 0, 4, 2, 6};
    const size_t stride = (size_t)width * channels;
    std::vector<uint8_t> raw((stride + 1) * (size_t)height);
    for (int y = 0; y < height; ++y) {
        const uint8_t *src = pixels + stride * (size_t)(flipY ? height - 1 - y : y);
        raw[(stride + 1) * (size_t)y] = 0;   // filter: none
        std::memcpy(&raw[(stride + 1) * (size_t)y + 1], src, stride);
    }
    uLongf zlen = compressBound((uLong)raw.size());
    std::vector<uint8_t> z(zlen);
    if (compress2(z.data(), &zlen, raw.data(), (uLong)raw.size(), 6) != Z_OK) return RT_ERR_IO;
    FILE *f = std::fopen(path, "wb");
    if (!f) return RT_ERR_IO;
    auto put32 = [](uint8_t *p, uint32_t v) { p[0] = (uint8_t)(v >> 24); p[1] = (uint8_t)(v >> 16); p[2] = (uint8_t)(v >> 8); p[3] = (uint8_t)v; };
    auto chunk = [&](const char *type, const uint8_t *data, size_t len) {
        uint8_t hdr[8];
        put32(hdr, (uint32_t)len);
        std::memcpy(hdr + 4, type, 4);
        std::fwrite(hdr, 1, 8, f);
        if (len) std::fwrite(data, 1, len, f);
        uLong crc = crc32(0L, (const Bytef *)type, 4);
        if (len) crc = crc32(crc, data, (uInt)len);
        uint8_t c[4];
        put32(c, (uint32_t)crc);
        std::fwrite(c, 1, 4, f);
    };
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    std::fwrite(sig, 1, 8, f);
    uint8_t ihdr[13];
    put32(ihdr, (uint32_t)width); put32(ihdr + 4, (uint32_t)height);
    ihdr[8] = 8; ihdr[9] = ctype[channels]; ihdr[10] = 0; ihdr[11] = 0; ihdr[12] = 0;
    chunk("IHDR", ihdr, 13);
    chunk("IDAT", z.data(), zlen);
    chunk("IEND", nullptr, 0);
    const bool ok = std::fclose(f) == 0;
    return ok ? RT_OK : RT_ERR_IO;
    });
}

int rt_cubemap_from_cross(const uint8_t *img, int width, int height, int channels, uint8_t *faces) {
    if (!img || !faces || channels < 1) return 0;
    if ((height % 3) != 0 || (width % 4) != 0 || (width / 4) != (height / 3)) return 0;   // cubemap.cpp:47
    const int n = height / 3;
    // cross cells (column,row) of +X -X +Y -Y +Z -Z, cubemap.cpp:86-91
    static const int cell[6][2] = {{2, 1}, {0, 1}, {1, 0}, {1, 2}, {1, 1}, {3, 1}};
    const size_t rowBytes = (size_t)n * channels, stride = (size_t)width * channels;
    for (int f = 0; f < 6; ++f) {
        const uint8_t *src = img + (size_t)cell[f][1] * n * stride + (size_t)cell[f][0] * rowBytes;
        uint8_t *dst = faces + (size_t)f * n * rowBytes;
        for (int y = 0; y < n; ++y) std::memcpy(dst + (size_t)y * rowBytes, src + (size_t)y * stride, rowBytes);
    }
    return n;
}

}  // extern "C"

// rt_present.hip -- the present pass of renderRay (src/render/render.cpp:199-239): shaders/rt/rt_present.frag
// restated as one HIP kernel: SVGF-lite 7x7 edge-aware filter (:126-225), ACES tonemap (:65-69), gamma 1/2.2 (:263),
// motion visualisation (:92-104).  Reads the four targets of the frame just rendered straight from their
// tile-major layout (NEAREST / CLAMP_TO_EDGE, as the GL textures are set up) and writes the RGBA8 "back buffer".
#include "rt_frame.hpp"

#pragma clang fp contract(off)

using namespace rtd;

namespace {

struct PresentTex {
    const uint2 *color;
    const uint32_t *motion;
    const uint2 *gpos, *gnrm;
    FrameGeom g;
    int blockSlots;   // > 0: the four targets are rank-major arrays of gathered blocks (tile-parallel frame on the gathering rank)
};

RT_DEV int texel_slot(const PresentTex &T, float u, float v) {   // texture(sampler2D, uv) with NEAREST + CLAMP_TO_EDGE
    const FrameGeom &g = T.g;
    int x = (int)__builtin_floorf(u * (float)g.W), y = (int)__builtin_floorf(v * (float)g.H);
    x = min(max(x, 0), g.W - 1);
    y = min(max(y, 0), g.H - 1);
    return T.blockSlots > 0 ? slot_in_gathered(g, x, y, T.blockSlots) : slot_of_pixel(g, x, y);
}
RT_DEV V3 aces(V3 x, float exposure) {
    x = x * exposure;
    const float a = 2.51f, b = 0.03f, c = 2.43f, d = 0.59f, e = 0.14f;
    V3 num = x * (a * x + mk3(b));
    V3 den = x * (c * x + mk3(d)) + mk3(e);
    return mk3(clampr(num.x / den.x, 0.0f, 1.0f), clampr(num.y / den.y, 0.0f, 1.0f), clampr(num.z / den.z, 0.0f, 1.0f));
}
RT_DEV V3 hsv2rgb(V3 c) {
    V3 q = mk3(fractr(c.x + 0.0f), fractr(c.x + 2.0f / 3.0f), fractr(c.x + 1.0f / 3.0f));
    V3 p = mk3(__builtin_fabsf(q.x * 6.0f - 3.0f), __builtin_fabsf(q.y * 6.0f - 3.0f), __builtin_fabsf(q.z * 6.0f - 3.0f));
    V3 k = mk3(clampr(p.x - 1.0f, 0.0f, 1.0f), clampr(p.y - 1.0f, 0.0f, 1.0f), clampr(p.z - 1.0f, 0.0f, 1.0f));
    return c.z * mix(mk3(1.0f), k, c.y);
}
RT_DEV uint32_t unorm8(float x) {
    float c = clampr(x, 0.0f, 1.0f);
    if (c != c) c = 0.0f;
    return (uint32_t)__builtin_rintf(c * 255.0f);
}
RT_DEV V2 unpack_half2(uint32_t r) { return mk2(f16_bits_to_f32((uint16_t)(r & 0xffffu)), f16_bits_to_f32((uint16_t)(r >> 16))); }

__global__ __launch_bounds__(256) void k_present(PresentTex T, RtPresentParams P, uint32_t *out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= T.g.W * T.g.H) return;
    const int px = i % T.g.W, py = i / T.g.W;
    const float u = (((float)px + 0.5f) + 0.5f) / (float)T.g.W, v = (((float)py + 0.5f) + 0.5f) / (float)T.g.H;   // rt_present.frag:233
    const int sc = texel_slot(T, u, v);
    V3 rgb;
    if (P.showMotion == 1) {
        V2 m = unpack_half2(T.motion[sc]);
        m = mk2(m.x * P.motionScale, m.y * P.motionScale);
        float mag = length(m);
        if (mag < 1e-4f) rgb = mk3(0.0f);
        else {
            float hue = atan2r(m.y, m.x) / (2.0f * 3.1415926535f) + 0.5f;
            rgb = hsv2rgb(mk3(hue, 1.0f, clampr(mag, 0.0f, 1.0f)));
        }
    } else {
        V4 rawc = unpack_half4(T.color[sc]);
        V3 raw = mk3(rawc.x, rawc.y, rawc.z);
        V3 linearColor = raw;
        if (P.enableSVGF != 0) {
            const V3 Y = mk3(0.299f, 0.587f, 0.114f);
            V3 cCenter = raw;
            float lCenter = dot(cCenter, Y);
            float varCenter = fminr(fmaxr(rawc.w - lCenter * lCenter, 0.0f), P.varMax);
            float motMag = length(unpack_half2(T.motion[sc]));
            V4 pc = unpack_half4(T.gpos[sc]), nc = unpack_half4(T.gnrm[sc]);
            V3 pCenter = mk3(pc.x, pc.y, pc.z), nCenter = mk3(nc.x, nc.y, nc.z);
            const float texelX = 1.0f / P.resolution[0], texelY = 1.0f / P.resolution[1];
            float t = clampr(smoothstepr(0.005f, 0.05f, motMag), 0.0f, 1.0f);
            float kVar = mixr(P.kVar, P.kVarMotion, t);
            float kColor = mixr(P.kColor, P.kColorMotion, t);
            const float K_NRM = 2.0f, K_POS = 0.02f;
            float varBoost = 1.0f + varCenter * (1.0f + kVar * 0.5f);
            V3 accumCol = mk3(0.0f);
            float accumW = 0.0f;
            for (int j = -3; j <= 3; ++j)
                for (int k = -3; k <= 3; ++k) {
                    float un = u + (float)k * texelX, vn = v + (float)j * texelY;
                    if (un < 0.0f || un > 1.0f || vn < 0.0f || vn > 1.0f) continue;
                    const int sn = texel_slot(T, un, vn);
                    V4 s = unpack_half4(T.color[sn]);
                    V3 c = mk3(s.x, s.y, s.z);
                    V3 dc = c - cCenter;
                    float wCol = expr(-dot(dc, dc) * (kColor * 0.3f + 0.05f));
                    V4 p4 = unpack_half4(T.gpos[sn]), n4 = unpack_half4(T.gnrm[sn]);
                    V3 dp = mk3(p4.x, p4.y, p4.z) - pCenter;
                    float wPos = expr(-dot(dp, dp) * K_POS);
                    float ndot = clampr(dot(normalize(nCenter), normalize(mk3(n4.x, n4.y, n4.z))), -1.0f, 1.0f);
                    float wNrm = expr(-fmaxr(0.0f, 1.0f - ndot) * K_NRM);
                    float wSpatial = (k == 0 && j == 0) ? 1.0f : 1.0f + varCenter * 4.0f;
                    float w = varBoost * wCol * wPos * wNrm * wSpatial;
                    accumCol = accumCol + c * w;
                    accumW = accumW + w;
                }
            V3 filtered = (accumW <= 0.0f) ? cCenter : accumCol / accumW;
            linearColor = mix(raw, filtered, clampr(P.svgfStrength, 0.0f, 1.0f));
        }
        V3 mapped = aces(linearColor, P.exposure);
        rgb = mk3(powr(mapped.x, 1.0f / 2.2f), powr(mapped.y, 1.0f / 2.2f), powr(mapped.z, 1.0f / 2.2f));
    }
    out[i] = unorm8(rgb.x) | (unorm8(rgb.y) << 8) | (unorm8(rgb.z) << 16) | (255u << 24);
}

}  // namespace

namespace rtl {
hipError_t launch_present(hipStream_t s, const FrameGeom &g, const uint2 *color, const uint32_t *motion, const uint2 *gpos,
                          const uint2 *gnrm, const RtPresentParams &p, uint32_t *outRGBA8, int gatheredBlockSlots) {
    PresentTex T;
    T.color = color; T.motion = motion; T.gpos = gpos; T.gnrm = gnrm; T.g = g; T.blockSlots = gatheredBlockSlots;
    const unsigned n = (unsigned)(g.W * g.H);
    hipLaunchKernelGGL(k_present, dim3((n + 255) / 256), dim3(256), 0, s, T, p, outRGBA8);
    return hipGetLastError();
}
}  // namespace rtl

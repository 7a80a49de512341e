// rt_device_math.hpp -- gfx950 device float model of the ray-trace path.
//
// The GLSL of shaders/rt/* leaves normalize/pow/sin/cos precision and a*b+c contraction to the GL
// driver.  This build fixes one model (DESIGN.md "Float model") and keeps it bit-reproducible:
// IEEE binary32 RNE for + - * / sqrt (hipcc's default correctly-rounded divide/sqrt), no implicit
// contraction (-ffp-contract=off and the pragma below), fused multiply-add only where written,
// v_min_f32 / v_max_f32 NaN semantics for min/max, and polynomial sin/cos/exp2/log2 with
// pow(x,y) = exp2(y*log2(x)) -- the way GPUs evaluate GLSL pow.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#pragma clang fp contract(off)

namespace rtd {

#define RT_DEV __device__ __forceinline__

struct V2 { float x, y; };
struct V3 { float x, y, z; };
struct V4 { float x, y, z, w; };

RT_DEV V3 mk3(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
RT_DEV V3 mk3(float s) { return mk3(s, s, s); }
RT_DEV V3 ld3(const float *p) { return mk3(p[0], p[1], p[2]); }
RT_DEV V2 mk2(float x, float y) { V2 r; r.x = x; r.y = y; return r; }
RT_DEV V4 mk4(float x, float y, float z, float w) { V4 r; r.x = x; r.y = y; r.z = z; r.w = w; return r; }
RT_DEV V3 operator+(V3 a, V3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
RT_DEV V3 operator-(V3 a, V3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
RT_DEV V3 operator-(V3 a) { return mk3(-a.x, -a.y, -a.z); }
RT_DEV V3 operator*(V3 a, V3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
RT_DEV V3 operator*(V3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
RT_DEV V3 operator*(float s, V3 a) { return mk3(s * a.x, s * a.y, s * a.z); }
RT_DEV V3 operator/(V3 a, float s) { return mk3(a.x / s, a.y / s, a.z / s); }

RT_DEV float fminr(float a, float b) { return __builtin_fminf(a, b); }
RT_DEV float fmaxr(float a, float b) { return __builtin_fmaxf(a, b); }
RT_DEV float clampr(float x, float lo, float hi) { return fminr(fmaxr(x, lo), hi); }
RT_DEV float mixr(float x, float y, float a) { return x * (1.0f - a) + y * a; }
RT_DEV float fractr(float x) { return x - __builtin_floorf(x); }
RT_DEV float smoothstepr(float e0, float e1, float x) {
    float t = clampr((x - e0) / (e1 - e0), 0.0f, 1.0f);
    return t * t * (3.0f - 2.0f * t);
}
RT_DEV float dot(V3 a, V3 b) { return __builtin_fmaf(a.z, b.z, __builtin_fmaf(a.y, b.y, a.x * b.x)); }
RT_DEV float dot(V2 a, V2 b) { return __builtin_fmaf(a.y, b.y, a.x * b.x); }
RT_DEV V3 cross(V3 a, V3 b) {
    return mk3(__builtin_fmaf(a.y, b.z, -(a.z * b.y)), __builtin_fmaf(a.z, b.x, -(a.x * b.z)),
               __builtin_fmaf(a.x, b.y, -(a.y * b.x)));
}
RT_DEV float length(V3 a) { return __builtin_sqrtf(dot(a, a)); }
RT_DEV float length(V2 a) { return __builtin_sqrtf(dot(a, a)); }
RT_DEV V3 normalize(V3 a) { float inv = 1.0f / __builtin_sqrtf(dot(a, a)); return a * inv; }
RT_DEV V3 mix(V3 x, V3 y, float a) { return mk3(mixr(x.x, y.x, a), mixr(x.y, y.y, a), mixr(x.z, y.z, a)); }
RT_DEV V3 reflect(V3 I, V3 N) { float k = 2.0f * dot(N, I); return I - k * N; }
RT_DEV V3 refract(V3 I, V3 N, float eta) {
    float d = dot(N, I);
    float k = 1.0f - eta * eta * (1.0f - d * d);
    if (k < 0.0f) return mk3(0.0f);
    return eta * I - (eta * d + __builtin_sqrtf(k)) * N;
}

RT_DEV uint32_t f2u(float f) { return __builtin_bit_cast(uint32_t, f); }
RT_DEV float u2f(uint32_t u) { return __builtin_bit_cast(float, u); }

// sin & cos together: Cody-Waite reduction by pi/2 (three fused steps), degree-7/8 polynomials.
RT_DEV void sincosr(float x, float &s, float &c) {
    float kf = __builtin_rintf(x * 0x1.45f306p-1f);
    float r = __builtin_fmaf(kf, -0x1.921fb6p+0f, x);
    r = __builtin_fmaf(kf, 0x1.777a5cp-25f, r);
    r = __builtin_fmaf(kf, 0x1.ee59dap-50f, r);
    int q = (int)kf & 3;
    float z = r * r;
    float ps = __builtin_fmaf(__builtin_fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f);
    float sn = __builtin_fmaf(ps * z, r, r);
    float pc = __builtin_fmaf(__builtin_fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f);
    float cs = __builtin_fmaf(pc * z, z, __builtin_fmaf(-0.5f, z, 1.0f));
    float so = (q & 1) ? cs : sn;
    float co = (q & 1) ? sn : cs;
    if (q == 1 || q == 2) co = -co;
    if (q >= 2) so = -so;
    s = so;
    c = co;
}
RT_DEV float log2r(float x) {
    int e = 0;
    uint32_t u = f2u(x);
    if (u < 0x00800000u) { x = x * 0x1p24f; u = f2u(x); e = -24; }
    e += (int)(u >> 23) - 127;
    float m = u2f((u & 0x007fffffu) | 0x3f800000u);
    if (m > 0x1.6a09e6p+0f) { m = m * 0.5f; e += 1; }
    float z = (m - 1.0f) / (m + 1.0f);
    float w = z * z;
    float p = __builtin_fmaf(w, 0x1.c71c72p-4f, 0x1.24924ap-3f);
    p = __builtin_fmaf(p, w, 0x1.99999ap-3f);
    p = __builtin_fmaf(p, w, 0x1.555556p-2f);
    p = __builtin_fmaf(p, w, 1.0f);
    float lnm = 2.0f * z * p;
    return __builtin_fmaf(lnm, 0x1.715476p+0f, (float)e);
}
RT_DEV float exp2r(float t) {
    if (t != t) return t;
    if (t > 128.0f) return __builtin_inff();
    if (t < -150.0f) return 0.0f;
    float nf = __builtin_rintf(t);
    float g = (t - nf) * 0x1.62e43p-1f;
    float p = __builtin_fmaf(g, 0x1.a01a02p-13f, 0x1.6c16c2p-10f);
    p = __builtin_fmaf(p, g, 0x1.111112p-7f);
    p = __builtin_fmaf(p, g, 0x1.555556p-5f);
    p = __builtin_fmaf(p, g, 0x1.555556p-3f);
    p = __builtin_fmaf(p, g, 0.5f);
    p = __builtin_fmaf(p, g, 1.0f);
    p = __builtin_fmaf(p, g, 1.0f);
    int n = (int)nf;
    int n1 = n / 2, n2 = n - n1;
    float s1 = u2f((uint32_t)(n1 + 127) << 23);
    float s2 = u2f((uint32_t)(n2 + 127) << 23);
    return p * s1 * s2;
}
RT_DEV float powr(float x, float y) {
    if (x < 0.0f || x != x) return __builtin_nanf("");
    if (x == 0.0f) return (y > 0.0f) ? 0.0f : ((y == 0.0f) ? 1.0f : __builtin_inff());
    if (x == __builtin_inff()) return (y > 0.0f) ? __builtin_inff() : ((y == 0.0f) ? 1.0f : 0.0f);
    return exp2r(y * log2r(x));
}

RT_DEV float expr(float x) { return exp2r(x * 0x1.715476p+0f); }
RT_DEV float atanpolyr(float z) {
    float w = z * z;
    float p = __builtin_fmaf(w, 0.0028662257f, -0.0161657367f);
    p = __builtin_fmaf(p, w, 0.0429096138f);
    p = __builtin_fmaf(p, w, -0.0752896400f);
    p = __builtin_fmaf(p, w, 0.1065626393f);
    p = __builtin_fmaf(p, w, -0.1420889944f);
    p = __builtin_fmaf(p, w, 0.1999355085f);
    p = __builtin_fmaf(p, w, -0.3333314528f);
    p = __builtin_fmaf(p, w, 1.0f);
    return z * p;
}
RT_DEV float atan2r(float y, float x) {
    const float PI = 3.14159265358979f, PIO2 = 1.57079632679490f;
    float ax = __builtin_fabsf(x), ay = __builtin_fabsf(y);
    if (ax == 0.0f && ay == 0.0f) return 0.0f;
    float a = (ay > ax) ? PIO2 - atanpolyr(ax / ay) : atanpolyr(ay / ax);
    if (x < 0.0f) a = PI - a;
    return (y < 0.0f) ? -a : a;
}

// binary32 -> binary16 bit pattern, round-to-nearest-even (v_cvt_f16_f32).
RT_DEV uint16_t f32_to_f16_bits(float f) {
    _Float16 h = (_Float16)f;
    return __builtin_bit_cast(uint16_t, h);
}
RT_DEV float f16_bits_to_f32(uint16_t b) {
    _Float16 h = __builtin_bit_cast(_Float16, b);
    return (float)h;
}

}  // namespace rtd

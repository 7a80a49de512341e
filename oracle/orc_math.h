// oracle/orc_math.h -- TEST INFRASTRUCTURE ONLY (parity oracle). Never linked into the product.
//
// Scalar fp32 "float model" used by the CPU restatement of shaders/rt/*.glsl.
// The GLSL spec leaves the precision of normalize/pow/sin/cos and the contraction of a*b+c to
// the driver ("not pinnable at all"), so the oracle fixes ONE model and the
// HIP path has to reproduce it bit for bit:
//   * every + - * / sqrt is IEEE-754 binary32, round-to-nearest-even, no implicit contraction
//     (compile with -ffp-contract=off); fused multiply-adds appear only where written (fmaf);
//   * dot()/cross() are the mul+fma chains a GPU compiler emits for GLSL dot/cross;
//   * min/max ignore a NaN operand (IEEE minNum/maxNum, the behaviour of v_min_f32/v_max_f32);
//   * sin/cos/exp2/log2/pow are the polynomial routines below (<= 2 ulp vs libm on the ranges
//     the shaders use; checked by tests/test_oracle_math.py), pow(x,y) = exp2(y*log2(x)) as
//     GPUs evaluate GLSL pow.
// Pins: the reference ships no golden vectors; the KATs in SURVEY.md 8c are checked by
// tests/test_oracle_kat.py, and the frames / rays the reference's own GLSL produced on SwiftShader by tests/test_glsl_reference.py
// (to a tolerance: that driver's transcendentals are its own).  The precision a desktop GL driver gives these functions is unpinnable.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

namespace orc {

static inline uint32_t f2u(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }

static inline float fmin_(float a, float b) { return std::fmin(a, b); }  // NaN-ignoring
static inline float fmax_(float a, float b) { return std::fmax(a, b); }
static inline float clampf(float x, float lo, float hi) { return fmin_(fmax_(x, lo), hi); }
static inline float mixf(float x, float y, float a) { return x * (1.0f - a) + y * a; }
static inline float fractf(float x) { return x - std::floor(x); }
static inline float smoothstepf(float e0, float e1, float x) {
    float t = clampf((x - e0) / (e1 - e0), 0.0f, 1.0f);
    return t * t * (3.0f - 2.0f * t);
}

struct vec2 { float x, y; };
struct vec3 { float x, y, z; };

static inline vec3 v3(float x, float y, float z) { return vec3{x, y, z}; }
static inline vec3 v3(float s) { return vec3{s, s, s}; }
static inline vec3 operator+(vec3 a, vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
static inline vec3 operator-(vec3 a, vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
static inline vec3 operator-(vec3 a) { return {-a.x, -a.y, -a.z}; }
static inline vec3 operator*(vec3 a, vec3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
static inline vec3 operator*(vec3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
static inline vec3 operator*(float s, vec3 a) { return {s * a.x, s * a.y, s * a.z}; }
static inline vec3 operator/(vec3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
static inline vec3 &operator+=(vec3 &a, vec3 b) { a = a + b; return a; }
static inline vec3 &operator*=(vec3 &a, vec3 b) { a = a * b; return a; }
static inline vec3 &operator*=(vec3 &a, float s) { a = a * s; return a; }

static inline float dot(vec3 a, vec3 b) { return std::fmaf(a.z, b.z, std::fmaf(a.y, b.y, a.x * b.x)); }
static inline float dot(vec2 a, vec2 b) { return std::fmaf(a.y, b.y, a.x * b.x); }
static inline vec3 cross(vec3 a, vec3 b) {
    return {std::fmaf(a.y, b.z, -(a.z * b.y)),
            std::fmaf(a.z, b.x, -(a.x * b.z)),
            std::fmaf(a.x, b.y, -(a.y * b.x))};
}
static inline float length(vec3 a) { return std::sqrt(dot(a, a)); }
static inline float length(vec2 a) { return std::sqrt(dot(a, a)); }
static inline vec3 normalize(vec3 a) { float inv = 1.0f / std::sqrt(dot(a, a)); return a * inv; }
static inline vec3 mix(vec3 x, vec3 y, float a) { return {mixf(x.x, y.x, a), mixf(x.y, y.y, a), mixf(x.z, y.z, a)}; }
static inline vec3 reflect(vec3 I, vec3 N) { float k = 2.0f * dot(N, I); return I - k * N; }
static inline vec3 refract(vec3 I, vec3 N, float eta) {
    float d = dot(N, I);
    float k = 1.0f - eta * eta * (1.0f - d * d);
    if (k < 0.0f) return v3(0.0f);
    return eta * I - (eta * d + std::sqrt(k)) * N;
}

// ---- deterministic transcendental set ------------------------------------------------------
// sin/cos: k = rint(x*2/pi); r = x - k*pi/2 (3-term Cody-Waite, fused); degree-7/8 minimax on
// [-pi/4, pi/4] (the classic single-precision Cephes coefficients).
static inline void sincos_core(float x, float *s, float *c) {
    float kf = std::rint(x * 0x1.45f306p-1f);            // 2/pi
    float r = std::fmaf(kf, -0x1.921fb6p+0f, x);         // pi/2 hi
    r = std::fmaf(kf, 0x1.777a5cp-25f, r);               // -(pi/2 mid) = +4.371139e-08
    r = std::fmaf(kf, 0x1.ee59dap-50f, r);               // -(pi/2 lo)  = +1.7151245e-15
    int q = (int)kf & 3;
    float z = r * r;
    float ps = std::fmaf(std::fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f);
    float sn = std::fmaf(ps * z, r, r);
    float pc = std::fmaf(std::fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f);
    float cs = std::fmaf(pc * z, z, std::fmaf(-0.5f, z, 1.0f));
    float so = (q & 1) ? cs : sn;
    float co = (q & 1) ? sn : cs;
    if (q == 1 || q == 2) co = -co;
    if (q >= 2) so = -so;
    *s = so; *c = co;
}
static inline float sinf_(float x) { float s, c; sincos_core(x, &s, &c); return s; }
static inline float cosf_(float x) { float s, c; sincos_core(x, &s, &c); return c; }

// log2 for finite x > 0: x = m*2^e, m in [sqrt(.5), sqrt(2)); ln m = 2 atanh((m-1)/(m+1)).
static inline float log2f_(float x) {
    int e = 0;
    uint32_t u = f2u(x);
    if (u < 0x00800000u) { x = x * 0x1p24f; u = f2u(x); e = -24; }   // subnormal
    e += (int)(u >> 23) - 127;
    uint32_t mant = (u & 0x007fffffu) | 0x3f800000u;                  // [1,2)
    float m = u2f(mant);
    if (m > 0x1.6a09e6p+0f) { m = m * 0.5f; e += 1; }                 // > sqrt(2)
    float z = (m - 1.0f) / (m + 1.0f);
    float w = z * z;
    float p = std::fmaf(w, 0x1.c71c72p-4f, 0x1.24924ap-3f);           // 1/9, 1/7
    p = std::fmaf(p, w, 0x1.99999ap-3f);                              // 1/5
    p = std::fmaf(p, w, 0x1.555556p-2f);                              // 1/3
    p = std::fmaf(p, w, 1.0f);
    float lnm = 2.0f * z * p;
    return std::fmaf(lnm, 0x1.715476p+0f, (float)e);                  // * 1/ln2 + e
}
// exp2 for any finite t.
static inline float exp2f_(float t) {
    if (t != t) return t;
    if (t > 128.0f) return INFINITY;
    if (t < -150.0f) return 0.0f;
    float nf = std::rint(t);
    float g = (t - nf) * 0x1.62e43p-1f;                               // * ln2, |g| <= .3466
    float p = std::fmaf(g, 0x1.a01a02p-13f, 0x1.6c16c2p-10f);        // 1/5040, 1/720
    p = std::fmaf(p, g, 0x1.111112p-7f);                              // 1/120
    p = std::fmaf(p, g, 0x1.555556p-5f);                              // 1/24
    p = std::fmaf(p, g, 0x1.555556p-3f);                              // 1/6
    p = std::fmaf(p, g, 0.5f);
    p = std::fmaf(p, g, 1.0f);
    p = std::fmaf(p, g, 1.0f);
    int n = (int)nf;
    int n1 = n / 2, n2 = n - n1;                                      // both within [-75, 64]
    float s1 = u2f((uint32_t)(n1 + 127) << 23);
    float s2 = u2f((uint32_t)(n2 + 127) << 23);
    return p * s1 * s2;
}
// GLSL pow(x,y) as GPUs evaluate it: exp2(y*log2(x)); x<0 undefined -> NaN.
static inline float powf_(float x, float y) {
    if (x < 0.0f || x != x) return NAN;
    if (x == 0.0f) return (y > 0.0f) ? 0.0f : ((y == 0.0f) ? 1.0f : INFINITY);
    if (x == INFINITY) return (y > 0.0f) ? INFINITY : ((y == 0.0f) ? 1.0f : 0.0f);
    return exp2f_(y * log2f_(x));
}

// exp(x) = exp2(x * log2(e)); atan2 through the 8-term A&S 4.4.49 odd polynomial on [-1, 1] (present pass only)
static inline float expf_(float x) { return exp2f_(x * 0x1.715476p+0f); }
static inline float atanpoly_(float z) {   // |z| <= 1
    float w = z * z;
    float p = std::fmaf(w, 0.0028662257f, -0.0161657367f);
    p = std::fmaf(p, w, 0.0429096138f);
    p = std::fmaf(p, w, -0.0752896400f);
    p = std::fmaf(p, w, 0.1065626393f);
    p = std::fmaf(p, w, -0.1420889944f);
    p = std::fmaf(p, w, 0.1999355085f);
    p = std::fmaf(p, w, -0.3333314528f);
    p = std::fmaf(p, w, 1.0f);
    return z * p;
}
static inline float atan2f_(float y, float x) {
    const float PI = 3.14159265358979f, PIO2 = 1.57079632679490f;
    float ax = std::fabs(x), ay = std::fabs(y);
    if (ax == 0.0f && ay == 0.0f) return 0.0f;
    float a = (ay > ax) ? PIO2 - atanpoly_(ax / ay) : atanpoly_(ay / ax);
    if (x < 0.0f) a = PI - a;
    return (y < 0.0f) ? -a : a;
}

// ---- fp16 (IEEE binary16) round-to-nearest-even, the conversion GL applies on RGBA16F stores --
static inline uint16_t f32_to_f16(float f) {
    uint32_t x = f2u(f);
    uint32_t sign = (x >> 16) & 0x8000u;
    uint32_t ax = x & 0x7fffffffu;
    if (ax >= 0x7f800000u) return (uint16_t)(sign | 0x7c00u | ((ax > 0x7f800000u) ? (0x0200u | ((ax >> 13) & 0x3ffu)) : 0u));
    if (ax >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);         // rounds to >= 65520 -> inf
    if (ax < 0x33000001u) return (uint16_t)sign;                      // <= 2^-25 -> 0
    int e = (int)(ax >> 23) - 127;
    uint32_t m = (ax & 0x007fffffu) | 0x00800000u;
    if (e < -14) {                                                    // subnormal half
        int sh = -14 - e + 13;                                        // 14..24
        uint32_t r = m >> sh;
        uint32_t rem = m & ((1u << sh) - 1u);
        uint32_t half = 1u << (sh - 1);
        if (rem > half || (rem == half && (r & 1u))) r++;
        return (uint16_t)(sign | r);
    }
    uint32_t r = ((uint32_t)(e + 15) << 10) | ((m >> 13) & 0x3ffu);
    uint32_t rem = m & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (r & 1u))) r++;
    return (uint16_t)(sign | r);
}
static inline float f16_to_f32(uint16_t h) {
    uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
    uint32_t e = (h >> 10) & 0x1fu, m = h & 0x3ffu;
    if (e == 0) {
        if (m == 0) return u2f(sign);
        float v = (float)m * 0x1p-24f;
        return sign ? -v : v;
    }
    if (e == 31) return u2f(sign | 0x7f800000u | (m << 13));
    return u2f(sign | ((e + 112u) << 23) | (m << 13));
}

}  // namespace orc

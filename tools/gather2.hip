// micro-benchmark, round 4: is the quad-cooperative record fetch worth its in-quad transpose in the regime the
// traversal kernels run in?  tools/gather.hip (round 2) compared the shapes with every line missing the L1 (1 MB table, uniformly
// random), all lanes on or whole quads off, and without moving the pieces to the lanes that need them.  Here:
//   * the table index is drawn from a HOT set (L1-resident) with probability hot/256, else from the whole table -> tunable L1 hit rate
//   * lanes are switched off by a 64-bit mask (scattered lanes, as after divergence), not by whole quads
//   * quad-mates stand on the SAME record with probability same/256 (the kernels' quad-merge factor is 0.68)
//   * shapes:  P  per-lane: 4 x dwordx4 of the lane's own 64-byte record                      (what k_trace does)
//              C  cooperative: instruction k = the quad fetches the record of quad-lane k, lane q piece q; no transpose (lower bound)
//              T  C + 4x4 in-quad transpose (DPP quad_perm + v_cndmask), every lane ends with its own record's four pieces
//              D  T, and quad-lanes whose record equals that of a lower quad-lane skip their fetch (dedup) and copy
//   hipcc --offload-arch=gfx950 -O3 -o gather2 tools/gather2.hip && ./gather2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float v4f __attribute__((ext_vector_type(4)));
#define DEV __device__ __forceinline__

template <int CTRL> DEV int dpp_i(int v) { return __builtin_amdgcn_mov_dpp(v, CTRL, 0xf, 0xf, true); }
template <int CTRL> DEV float dpp_f(float v) { return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xf, 0xf, true)); }
template <int CTRL> DEV v4f dpp_v(v4f v) { v4f r; r.x = dpp_f<CTRL>(v.x); r.y = dpp_f<CTRL>(v.y); r.z = dpp_f<CTRL>(v.z); r.w = dpp_f<CTRL>(v.w); return r; }
constexpr int QP_BCAST(int k) { return k * 0x55; }
constexpr int QP_XOR1 = 0xB1, QP_XOR2 = 0x4E;
DEV v4f sel(bool c, v4f a, v4f b) { v4f r; r.x = c ? a.x : b.x; r.y = c ? a.y : b.y; r.z = c ? a.z : b.z; r.w = c ? a.w : b.w; return r; }

// 4x4 transpose inside each lane quad: in: r[k] of lane q = piece q of record k; out: r[j] of lane q = piece j of record q
DEV void quad_transpose(v4f &r0, v4f &r1, v4f &r2, v4f &r3, uint32_t q) {
    const bool b0 = q & 1u, b1 = q & 2u;
    // stage 1: 2x2 blocks over (register bit 0, lane bit 0)
    v4f a0 = sel(b0, dpp_v<QP_XOR1>(r1), r0), a1 = sel(b0, r1, dpp_v<QP_XOR1>(r0));
    v4f a2 = sel(b0, dpp_v<QP_XOR1>(r3), r2), a3 = sel(b0, r3, dpp_v<QP_XOR1>(r2));
    // stage 2: (register bit 1, lane bit 1)
    r0 = sel(b1, dpp_v<QP_XOR2>(a2), a0); r2 = sel(b1, a2, dpp_v<QP_XOR2>(a0));
    r1 = sel(b1, dpp_v<QP_XOR2>(a3), a1); r3 = sel(b1, a3, dpp_v<QP_XOR2>(a1));
}

DEV void pin(v4f &v) { asm volatile("" : "+v"(v)); }

// The same transpose with v_cndmask_b32_dpp (select + cross-lane read in ONE instruction; hipcc emits v_mov_b32_dpp + v_cndmask_b32 for the
// C++ form above: 64 vector instructions, here 32 + 8 scalar moves).  outA = keepA ? x : partner(y), outB = keepB ? y : partner(x).
// s_nop 1: a DPP source written by a VALU instruction needs two wait states, which the assembler does not insert inside inline asm.
#define RT_QT_BLOCK(PERM)                                                                                                                     \
    asm volatile("s_nop 1\n\ts_mov_b64 vcc, %[ka]\n\t"                                                                                       \
                 "v_cndmask_b32_dpp %[a0], %[y0], %[x0], vcc quad_perm:" PERM " row_mask:0xf bank_mask:0xf\n\t"                                \
                 "v_cndmask_b32_dpp %[a1], %[y1], %[x1], vcc quad_perm:" PERM " row_mask:0xf bank_mask:0xf\n\t"                                \
                 "v_cndmask_b32_dpp %[a2], %[y2], %[x2], vcc quad_perm:" PERM " row_mask:0xf bank_mask:0xf\n\t"                                \
                 "v_cndmask_b32_dpp %[a3], %[y3], %[x3], vcc quad_perm:" PERM " row_mask:0xf bank_mask:0xf\n\t"                                \
                 "s_mov_b64 vcc, %[kb]\n\t"                                                                                                   \
                 "v_cndmask_b32_dpp %[b0], %[x0], %[y0], vcc quad_perm:" PERM " row_mask:0xf bank_mask:0xf\n\t"                                \
                 "v_cndmask_b32_dpp %[b1], %[x1], %[y1], vcc quad_perm:" PERM " row_mask:0xf bank_mask:0xf\n\t"                                \
                 "v_cndmask_b32_dpp %[b2], %[x2], %[y2], vcc quad_perm:" PERM " row_mask:0xf bank_mask:0xf\n\t"                                \
                 "v_cndmask_b32_dpp %[b3], %[x3], %[y3], vcc quad_perm:" PERM " row_mask:0xf bank_mask:0xf"                                     \
                 : [a0] "=&v"(oa.x), [a1] "=&v"(oa.y), [a2] "=&v"(oa.z), [a3] "=&v"(oa.w), [b0] "=&v"(ob.x), [b1] "=&v"(ob.y), [b2] "=&v"(ob.z), [b3] "=&v"(ob.w) \
                 : [x0] "v"(x.x), [x1] "v"(x.y), [x2] "v"(x.z), [x3] "v"(x.w), [y0] "v"(y.x), [y1] "v"(y.y), [y2] "v"(y.z), [y3] "v"(y.w), [ka] "s"(keepA), [kb] "s"(keepB) \
                 : "vcc")
DEV void qt_pair_xor1(v4f x, v4f y, v4f &oa, v4f &ob, unsigned long long keepA, unsigned long long keepB) { RT_QT_BLOCK("[1,0,3,2]"); }
DEV void qt_pair_xor2(v4f x, v4f y, v4f &oa, v4f &ob, unsigned long long keepA, unsigned long long keepB) { RT_QT_BLOCK("[2,3,0,1]"); }
DEV void quad_transpose_asm(v4f &r0, v4f &r1, v4f &r2, v4f &r3) {
    const unsigned long long EVEN = 0x5555555555555555ull, ODD = 0xAAAAAAAAAAAAAAAAull, LO = 0x3333333333333333ull, HI = 0xCCCCCCCCCCCCCCCCull;
    v4f a0, a1, a2, a3;
    qt_pair_xor1(r0, r1, a0, a1, EVEN, ODD);
    qt_pair_xor1(r2, r3, a2, a3, EVEN, ODD);
    qt_pair_xor2(a0, a2, r0, r2, LO, HI);
    qt_pair_xor2(a1, a3, r1, r3, LO, HI);
}

template <int MODE>
__global__ __launch_bounds__(256) void k(const v4f *__restrict__ tab, uint32_t nn, uint32_t hotN, int hot, int same, int steps, float *out, unsigned long long mask) {
    const uint32_t lane = threadIdx.x & 63, q = lane & 3u;
    uint32_t rng = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
    uint32_t idx = rng % nn;
    float acc = 0.f;
    const bool on = (mask >> lane) & 1ull;
    for (int s = 0; s < steps; ++s) {
        v4f a = {0, 0, 0, 0}, b = a, c = a, d = a;
        if (MODE == 0) {
            if (on) { const v4f *p = tab + (size_t)idx * 4; a = p[0]; b = p[1]; c = p[2]; d = p[3]; pin(a); pin(b); pin(c); pin(d); }
        } else {
            const int me = on ? (int)idx : -1;
            const int i0 = dpp_i<QP_BCAST(0)>(me), i1 = dpp_i<QP_BCAST(1)>(me), i2 = dpp_i<QP_BCAST(2)>(me), i3 = dpp_i<QP_BCAST(3)>(me);
            bool f0 = i0 >= 0, f1 = i1 >= 0, f2 = i2 >= 0, f3 = i3 >= 0;
            if (MODE == 3 || MODE == 5) { f1 = f1 && i1 != i0; f2 = f2 && i2 != i0 && i2 != i1; f3 = f3 && i3 != i0 && i3 != i1 && i3 != i2; }
            if (f0) a = tab[(size_t)i0 * 4 + q];
            if (f1) b = tab[(size_t)i1 * 4 + q];
            if (f2) c = tab[(size_t)i2 * 4 + q];
            if (f3) d = tab[(size_t)i3 * 4 + q];
            pin(a); pin(b); pin(c); pin(d);
            if (MODE == 3 || MODE == 5) {   // a skipped fetch copies the first equal record's pieces
                d = (i3 >= 0 && i3 == i0) ? a : (i3 >= 0 && i3 == i1) ? b : (i3 >= 0 && i3 == i2) ? c : d;
                c = (i2 >= 0 && i2 == i0) ? a : (i2 >= 0 && i2 == i1) ? b : c;
                b = (i1 >= 0 && i1 == i0) ? a : b;
            }
            if (MODE == 2 || MODE == 3) quad_transpose(a, b, c, d, q);
            if (MODE >= 4) quad_transpose_asm(a, b, c, d);
        }
        if (on) {
            // every dword is used (a node visit consumes the whole record), so nothing of the transpose can be optimised away
            acc += ((a.x + a.y) + (a.z + a.w)) + ((b.x + b.y) + (b.z + b.w)) + ((c.x + c.y) + (c.z + c.w)) + ((d.x + d.y) + (d.z + d.w));
            rng = rng * 1664525u + 1013904223u + (uint32_t)(int)acc;
            const uint32_t r = rng >> 8;
            idx = ((r & 255u) < (uint32_t)hot) ? (r >> 8) % hotN : (r >> 8) % nn;
        }
        // quad-mates on the same record with probability same/256: the lane copies the record index of its quad's lane 0
        const uint32_t lead = (uint32_t)dpp_i<QP_BCAST(0)>((int)idx);
        if (((rng >> 3) & 255u) < (uint32_t)same) idx = lead;
    }
    if (acc == 12345.f) out[0] = acc;
}

// correctness of the two transposes: every lane fetches a record of a table with distinct values three ways and stores its 16 floats
template <int MODE>
__global__ void k_verify(const v4f *__restrict__ tab, uint32_t nn, v4f *out, unsigned long long mask) {
    const uint32_t lane = threadIdx.x & 63, q = lane & 3u;
    const uint32_t idx = (threadIdx.x * 2654435761u >> 7) % nn;
    const bool on = (mask >> lane) & 1ull;
    v4f a = {0, 0, 0, 0}, b = a, c = a, d = a;
    if (MODE == 0) { if (on) { const v4f *p = tab + (size_t)idx * 4; a = p[0]; b = p[1]; c = p[2]; d = p[3]; } }
    else {
        const int me = on ? (int)idx : -1;
        const int i0 = dpp_i<QP_BCAST(0)>(me), i1 = dpp_i<QP_BCAST(1)>(me), i2 = dpp_i<QP_BCAST(2)>(me), i3 = dpp_i<QP_BCAST(3)>(me);
        if (i0 >= 0) a = tab[(size_t)i0 * 4 + q];
        if (i1 >= 0) b = tab[(size_t)i1 * 4 + q];
        if (i2 >= 0) c = tab[(size_t)i2 * 4 + q];
        if (i3 >= 0) d = tab[(size_t)i3 * 4 + q];
        if (MODE == 1) quad_transpose(a, b, c, d, q); else quad_transpose_asm(a, b, c, d);
    }
    if (on) { out[threadIdx.x * 4 + 0] = a; out[threadIdx.x * 4 + 1] = b; out[threadIdx.x * 4 + 2] = c; out[threadIdx.x * 4 + 3] = d; }
}
static int verify() {
    const uint32_t nn = 4096;
    std::vector<float> h((size_t)nn * 16);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)i;
    v4f *tab, *out; hipMalloc(&tab, h.size() * 4); hipMalloc(&out, 256 * 64 * 3);
    hipMemcpy(tab, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    int bad = 0;
    for (unsigned long long mask : {~0ull, 0x9E3779B97F4A7C15ull, 0x0101010101010101ull * 0x11ull}) {
        std::vector<float> r[3];
        for (int m = 0; m < 3; ++m) {
            hipMemset(out, 0, 256 * 64 * 3);
            if (m == 0) hipLaunchKernelGGL(k_verify<0>, dim3(1), dim3(256), 0, 0, tab, nn, out, mask);
            if (m == 1) hipLaunchKernelGGL(k_verify<1>, dim3(1), dim3(256), 0, 0, tab, nn, out, mask);
            if (m == 2) hipLaunchKernelGGL(k_verify<2>, dim3(1), dim3(256), 0, 0, tab, nn, out, mask);
            r[m].resize(256 * 16);
            hipMemcpy(r[m].data(), out, 256 * 64, hipMemcpyDeviceToHost);
        }
        for (int m = 1; m < 3; ++m) for (size_t i = 0; i < r[0].size(); ++i) if (r[m][i] != r[0][i]) bad++;
    }
    printf("verify: quad transposes (C++ / asm) vs per-lane fetch: %d mismatching floats\n", bad);
    hipFree(tab); hipFree(out);
    return bad;
}

static double run_one(int mode, const v4f *tab, uint32_t nn, uint32_t hotN, int hot, int same, float *out, int blocks, unsigned long long mask, int steps) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto go = [&]() {
        switch (mode) {
        case 0: hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, tab, nn, hotN, hot, same, steps, out, mask); break;
        case 1: hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, tab, nn, hotN, hot, same, steps, out, mask); break;
        case 2: hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, tab, nn, hotN, hot, same, steps, out, mask); break;
        case 3: hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), 0, 0, tab, nn, hotN, hot, same, steps, out, mask); break;
        case 4: hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(256), 0, 0, tab, nn, hotN, hot, same, steps, out, mask); break;
        default: hipLaunchKernelGGL(k<5>, dim3(blocks), dim3(256), 0, 0, tab, nn, hotN, hot, same, steps, out, mask); break;
        }
    };
    go();
    hipEventRecord(e0); go(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0); hipEventDestroy(e1);
    return ms;
}

int main(int argc, char **argv) {
    const int blocks = 256 * 5, steps = 400;
    const uint32_t tableKB[] = {16, 1024, 8192};          // L1-resident / L2-resident / beyond one XCD's L2
    float *out; hipMalloc(&out, 64);
    if (verify()) return 1;
    const unsigned long long masks[] = {~0ull, 0x5A3C96C3A5693C5Aull, 0x9E3779B97F4A7C15ull, 0x0101010101010101ull * 0x11ull};
    const char *mname[] = {"64 lanes", "32 lanes, 2 per quad", "scattered (golden)", "16 lanes, 1 per quad"};
    const char *names[] = {"P per-lane", "C coop, no transpose", "T coop + transpose (C++)", "D coop + dedup + transpose", "U coop + transpose (asm)", "E coop + dedup + asm"};
    for (uint32_t kb : tableKB) {
        const uint32_t nn = kb * 1024 / 64;
        v4f *tab; hipMalloc(&tab, (size_t)nn * 64 + 256); hipMemset(tab, 0, (size_t)nn * 64 + 256);
        for (int hot : {0, 128, 224}) {
            if (kb == 16 && hot) continue;
            for (int same : {0, 96}) {
                for (int mi = 0; mi < 4; ++mi) {
                    double base = 0;
                    for (int mode = 0; mode < 6; ++mode) {
                        const double ms = run_one(mode, tab, nn, 128, hot, same, out, blocks, masks[mi], steps);
                        if (mode == 0) base = ms;
                        const int lanesOn = __builtin_popcountll(masks[mi]);
                        printf("table %5u KB hot %3d/256 same %3d/256 | %-22s | %-28s %7.3f ms  %5.2f x P   %6.1f G lane-steps/s\n", kb, hot, same, mname[mi], names[mode], ms,
                               ms / base, (double)blocks * 4 * steps * lanesOn / (ms * 1e-3) / 1e9);
                    }
                }
            }
        }
        hipFree(tab);
    }
    return 0;
}

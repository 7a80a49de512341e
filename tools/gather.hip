// micro-benchmark: what a dependent per-lane gather step costs under full occupancy, by access shape
//   hipcc --offload-arch=gfx950 -O3 -o gather tools/gather.hip && ./gather
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int MODE>
__global__ __launch_bounds__(256) void k(const float4 *__restrict__ tab, uint32_t nn, int steps, float *out, uint32_t activeMask) {
    const uint32_t lane = threadIdx.x & 63;
    uint32_t idx = (blockIdx.x * 256u + threadIdx.x) * 2654435761u % nn;
    float acc = 0.f;
    const bool on = (activeMask >> (lane & 31)) & 1u;   // lanes switched off by the mask idle through the loop
    if (on) for (int s = 0; s < steps; ++s) {
        float4 a, b, c, d;
        if (MODE == 0) { const float4 *p = tab + (size_t)idx * 4; a = p[0]; b = p[1]; c = p[2]; d = p[3]; acc += a.x + b.y + c.z + d.w; }
        if (MODE == 1) { const float4 *p = tab + (size_t)(idx & ~1u) * 4; a = p[0]; b = p[1]; c = p[2]; d = p[3]; float4 e = p[4], f = p[5], g = p[6], h = p[7]; acc += a.x + b.y + c.z + d.w + e.x + f.y + g.z + h.w; }
        if (MODE == 2) {   // cooperative: 4 lanes share one 64-byte node, four instructions fetch 64 nodes: 16 distinct lines per instruction
            uint32_t base = __shfl(idx, 0, 64);   // not realistic addressing, only the access shape matters
            uint32_t n0 = (base + (lane >> 2) * 977u) % nn, n1 = (n0 + 131u) % nn, n2 = (n0 + 257u) % nn, n3 = (n0 + 389u) % nn;
            a = tab[(size_t)n0 * 4 + (lane & 3)]; b = tab[(size_t)n1 * 4 + (lane & 3)]; c = tab[(size_t)n2 * 4 + (lane & 3)]; d = tab[(size_t)n3 * 4 + (lane & 3)];
            acc += a.x + b.y + c.z + d.w;
        }
        if (MODE == 3) { const float4 *p = tab + (size_t)idx * 4; a = p[0]; b = p[1]; acc += a.x + b.y; }
        if (MODE == 4) { const float4 *p = tab + (size_t)idx * 4; a = p[0]; acc += a.x; }
        if (MODE == 5) { uint32_t u = __shfl(idx, 0, 64); const float4 *p = tab + (size_t)u * 4; a = p[0]; b = p[1]; c = p[2]; d = p[3]; acc += a.x + b.y + c.z + d.w; }
        if (MODE == 6) { uint32_t u = __shfl(idx, lane & ~3u, 64); const float4 *p = tab + (size_t)u * 4; a = p[0]; b = p[1]; c = p[2]; d = p[3]; acc += a.x + b.y + c.z + d.w; }   // 16 distinct nodes per wave
        idx = (idx * 1664525u + 1013904223u + (uint32_t)(int)acc) % nn;
    }
    if (acc == 12345.f) out[0] = acc;
}
template <int MODE> void run(const char *what, const float4 *tab, uint32_t nn, float *out, int blocks, uint32_t mask) {
    const int steps = 400;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, tab, nn, steps, out, mask);
    hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, tab, nn, steps, out, mask);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    int lanesOn = __builtin_popcount(mask) * 2;
    double waveSteps = (double)blocks * 4 * steps;
    printf("%-44s blocks %5d lanes/wave %2d: %7.3f ms  %6.1f ns per wave-step at %d waves/CU  -> %6.2f G lane-steps/s\n", what, blocks, lanesOn, ms,
           ms * 1e6 / steps, blocks * 4 / 256, waveSteps * lanesOn / (ms * 1e-3) / 1e9);
}
int main() {
    const uint32_t nn = 16384;   // 1 MB of 64-byte nodes: L2 resident, like the BVH
    float4 *tab; float *out;
    hipMalloc(&tab, (size_t)nn * 64 + 256); hipMemset(tab, 0, (size_t)nn * 64 + 256); hipMalloc(&out, 64);
    for (int blocks : {256 * 5}) for (uint32_t mask : {0xffffffffu, 0x0000ffffu, 0x000000ffu}) {
        run<0>("64 B per lane, 4 x dwordx4 (2-wide node)", tab, nn, out, blocks, mask);
        run<1>("128 B per lane, 8 x dwordx4 (4-wide node)", tab, nn, out, blocks, mask);
        run<3>("32 B per lane, 2 x dwordx4", tab, nn, out, blocks, mask);
        run<4>("16 B per lane, 1 x dwordx4", tab, nn, out, blocks, mask);
        run<2>("cooperative: 16 lines per instruction x 4", tab, nn, out, blocks, mask);
        run<6>("64 B per lane, 16 distinct nodes per wave", tab, nn, out, blocks, mask);
        run<5>("64 B per lane, all lanes the same node", tab, nn, out, blocks, mask);
    }
    return 0;
}

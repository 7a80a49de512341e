// micro-benchmark: shader clock vs wall clock in short kernels, and dependent-gather latency per step
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <random>
__global__ void chase(const float4* nodes, const int* next, int steps, int nnodes, unsigned long long* out, int stride4) {
    int lane = threadIdx.x + blockIdx.x * blockDim.x;
    int idx = (int)(((unsigned)lane * 7919u) % (unsigned)nnodes);
    unsigned long long c0 = clock64(), w0 = wall_clock64();
    float acc = 0.f;
    for (int s = 0; s < steps; ++s) {
        const float4* p = nodes + (size_t)idx * stride4;
        float4 a = p[0], b = p[1], c = p[2], d = p[3];
        acc += a.x + b.y + c.z + d.w;
        idx = next[idx];                 // dependent: next index from memory (like a child ref)
        idx = (idx + (int)(acc * 1e-30f)) % nnodes;
    }
    unsigned long long c1 = clock64(), w1 = wall_clock64();
    if (threadIdx.x == 0) { out[blockIdx.x * 3 + 0] = c1 - c0; out[blockIdx.x * 3 + 1] = w1 - w0; out[blockIdx.x * 3 + 2] = (unsigned long long)acc; }
}
int main() {
    const int nn = 16384, stride4 = 4;   // 1 MB of 64-byte nodes
    std::vector<int> nxt(nn); std::mt19937 r(1); for (auto& v : nxt) v = r() % nn;
    float4* dn; int* dx; unsigned long long* dout;
    hipMalloc(&dn, (size_t)nn * stride4 * 16); hipMemset(dn, 0, (size_t)nn * stride4 * 16);
    hipMalloc(&dx, nn * 4); hipMemcpy(dx, nxt.data(), nn * 4, hipMemcpyHostToDevice);
    hipMalloc(&dout, 4096 * 3 * 8);
    for (int blocks : {1, 64, 1280}) for (int rep = 0; rep < 3; ++rep) {
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipEventRecord(a);
        hipLaunchKernelGGL(chase, dim3(blocks), dim3(256), 0, 0, dn, dx, 200, nn, dout, stride4);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        unsigned long long h[3]; hipMemcpy(h, dout, 24, hipMemcpyDeviceToHost);
        printf("blocks=%4d rep=%d: kernel %.1f us; block0: %llu shader cycles, %llu wall ticks(100MHz) -> %.0f MHz; %.0f cycles/step = %.2f us/step\n", blocks, rep,
               ms * 1e3, h[0], h[1], h[1] ? (double)h[0] / h[1] * 100.0 : 0.0, h[0] / 200.0, h[1] / 100.0 / 200.0);
    }
    return 0;
}

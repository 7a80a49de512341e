// rt_bvh_gpu.hip -- the reference's median-split BVH (src/scene/bvh.cpp:41-135) built on the GPU ("GPU
// builder as an optional fast path"; the reference builds on the CPU only).
//
// The builder splits a range [begin,end) at mid = (begin+end)/2 and makes a leaf of <= 8 triangles (bvh.cpp:62,73), so the
// SHAPE of the tree -- node numbering (pre-order), ranges, leaf slots after the LIFO re-packing (:109-135) -- depends on
// the triangle count alone and is laid out on the host (skeleton()).  What depends on the data is done on the device,
// level by level, for all nodes of a depth at once:
//   * node bounds = min / max over the triangles of its range (bvh.cpp:49-54), block-reduced, one atomic per block and node;
//   * split axis = largest extent (:72);
//   * the partition about the median centroid (:75-80).  The reference uses std::nth_element; here every range is SORTED by
//     the same key (one global radix sort per level on (node rank, centroid[axis]) pairs).  The lower half holds the same
//     triangles whenever the median key is unique, so nodes, boxes and the set of triangles of every leaf are then identical
//     to the CPU build's; only the order of triangles inside a leaf differs (nth_element's arrangement is unspecified), which
//     can change the winner of an exact-t tie between two triangles of one leaf, nothing else.  rt_build_bvh stays the
//     parity path; this is the fast one (1 M triangles: ~0.9 s on the host).
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <vector>

#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>

#include "../../include/rt_mi355.h"

#pragma clang fp contract(off)

namespace {

struct SkNode { int begin, end, left, right, depth, firstOut; };

// pre-order numbering, exactly rt_host.cpp build_nodes / bvh.cpp build_recursive
void skeleton(int n, std::vector<SkNode> &nodes) {
    struct Work { int begin, end, parent, depth; bool isRight; };
    std::vector<Work> todo{{0, n, -1, 0, false}};
    while (!todo.empty()) {
        const Work w = todo.back();
        todo.pop_back();
        const int self = (int)nodes.size();
        nodes.push_back({w.begin, w.end, -1, -1, w.depth, -1});
        if (w.parent >= 0) (w.isRight ? nodes[(size_t)w.parent].right : nodes[(size_t)w.parent].left) = self;
        if (w.end - w.begin <= 8) continue;
        const int mid = (w.begin + w.end) / 2;
        todo.push_back({mid, w.end, self, w.depth + 1, true});
        todo.push_back({w.begin, mid, self, w.depth + 1, false});
    }
    // leaf re-packing: LIFO walk that pushes left then right (bvh.cpp:109-135)
    int out = 0;
    std::vector<int> walk{0};
    while (!walk.empty()) {
        const int i = walk.back();
        walk.pop_back();
        SkNode &nd = nodes[(size_t)i];
        if (nd.left < 0) { nd.firstOut = out; out += nd.end - nd.begin; }
        else { walk.push_back(nd.left); walk.push_back(nd.right); }
    }
}

__device__ __forceinline__ uint32_t f2sortable(float f) { uint32_t u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__device__ __forceinline__ float sortable2f(uint32_t s) { return __uint_as_float((s & 0x80000000u) ? (s & 0x7fffffffu) : ~s); }

// per triangle: bounds and centroid, the reference's expressions (bvh.cpp:10-26)
__global__ void k_tri_prep(const float *__restrict__ t9, int n, float *__restrict__ mn, float *__restrict__ mx, float *__restrict__ cen) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float *t = t9 + (size_t)i * 9;
    for (int a = 0; a < 3; ++a) {
        const float v0 = t[a], v1 = v0 + t[3 + a], v2 = v0 + t[6 + a];
        mn[(size_t)a * n + i] = fminf(v0, fminf(v1, v2));
        mx[(size_t)a * n + i] = fmaxf(v0, fmaxf(v1, v2));
        cen[(size_t)a * n + i] = ((v0 + v1) + v2) * (1.0f / 3.0f);
    }
}

// item position -> rank of the node of this level that contains it (levelBegin sorted ascending, binary search)
__device__ int find_seg(const int *__restrict__ segBegin, int nSeg, int pos) {
    int lo = 0, hi = nSeg - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (segBegin[mid] <= pos) lo = mid; else hi = mid - 1; }
    return lo;
}

// bounds of every node of one level: sortable-uint atomics, pre-reduced per wave when the whole wave lies in one node
__global__ void k_level_bounds(const int *__restrict__ perm, int n, const float *__restrict__ mn, const float *__restrict__ mx,
                               const int *__restrict__ segBegin, const int *__restrict__ segEnd, int nSeg, uint32_t *__restrict__ bounds /* [nSeg][6] */) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i < n;
    int seg = -1;
    if (live) { seg = find_seg(segBegin, nSeg, i); if (i < segBegin[seg] || i >= segEnd[seg]) seg = -1; }   // positions of finished leaves belong to no node of this level
    uint32_t v[6];
    for (int a = 0; a < 3; ++a) {
        const int t = live ? perm[i] : 0;
        v[a] = seg >= 0 ? f2sortable(mn[(size_t)a * n + t]) : 0xffffffffu;
        v[3 + a] = seg >= 0 ? f2sortable(mx[(size_t)a * n + t]) : 0u;
    }
    const int seg0 = __shfl(seg, 0, 64);
    const bool uniform = __ballot(seg != seg0) == 0ull;
    if (uniform) {
        if (seg0 < 0) return;
        for (int c = 0; c < 6; ++c) {
            uint32_t x = v[c];
            for (int off = 32; off > 0; off >>= 1) { const uint32_t y = __shfl_down(x, off, 64); x = c < 3 ? min(x, y) : max(x, y); }
            if ((threadIdx.x & 63) == 0) { if (c < 3) atomicMin(&bounds[(size_t)seg0 * 6 + c], x); else atomicMax(&bounds[(size_t)seg0 * 6 + c], x); }
        }
    } else if (seg >= 0) {
        for (int c = 0; c < 3; ++c) { atomicMin(&bounds[(size_t)seg * 6 + c], v[c]); atomicMax(&bounds[(size_t)seg * 6 + 3 + c], v[3 + c]); }
    }
}

// sort key of every item: (rank of its node at this level, centroid along that node's axis); items of nodes that are leaves
// at this level or were finished earlier keep their place (key = their position)
__global__ void k_level_keys(const int *__restrict__ perm, int n, const float *__restrict__ cen, const int *__restrict__ segBegin,
                             const int *__restrict__ segEnd, const int *__restrict__ segInner, int nSeg, const uint32_t *__restrict__ bounds,
                             unsigned long long *__restrict__ keys) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int seg = find_seg(segBegin, nSeg, i);
    // the high word orders ranges by where they start, so every range stays in place; the low word orders inside a range
    uint32_t lowKey = (uint32_t)i;
    uint32_t high = (uint32_t)i;                     // finished item: unique key = its own position
    if (i >= segBegin[seg] && i < segEnd[seg]) {
        high = (uint32_t)segBegin[seg];
        if (segInner[seg]) {
            const float ex = sortable2f(bounds[(size_t)seg * 6 + 3]) - sortable2f(bounds[(size_t)seg * 6 + 0]);
            const float ey = sortable2f(bounds[(size_t)seg * 6 + 4]) - sortable2f(bounds[(size_t)seg * 6 + 1]);
            const float ez = sortable2f(bounds[(size_t)seg * 6 + 5]) - sortable2f(bounds[(size_t)seg * 6 + 2]);
            const int axis = (ex > ey) ? ((ex > ez) ? 0 : 2) : ((ey > ez) ? 1 : 2);   // bvh.cpp:72
            lowKey = f2sortable(cen[(size_t)axis * n + perm[i]]);
        }
    }
    keys[i] = ((unsigned long long)high << 32) | lowKey;
}

__global__ void k_iota(int *p, int n) { const int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = i; }

__global__ void k_emit_tris(const float *__restrict__ t9, const int *__restrict__ perm, const int *__restrict__ outOfPos, int n, float *__restrict__ t12) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float *t = t9 + (size_t)perm[i] * 9;
    float *o = t12 + (size_t)outOfPos[i] * 12;
    o[0] = t[0]; o[1] = t[1]; o[2] = t[2]; o[3] = 0.0f;
    o[4] = t[3]; o[5] = t[4]; o[6] = t[5]; o[7] = 0.0f;
    o[8] = t[6]; o[9] = t[7]; o[10] = t[8]; o[11] = 0.0f;
}

#define BG_TRY(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { err = hipGetErrorString(e_); goto done; } } while (0)

}  // namespace

namespace rtl {

// -> number of nodes, or a negative RtStatus; *errOut names a HIP error
int build_bvh_gpu(int device, const float *tris9, int nTris, float *nodes12, float *tris12, const char **errOut) {
    if (nTris < 0 || (nTris > 0 && (!tris9 || !nodes12 || !tris12))) return RT_ERR_INVALID;
    if (nTris == 0) return 0;
    const char *err = nullptr;
    std::vector<SkNode> sk;
    skeleton(nTris, sk);
    int maxDepth = 0;
    for (const SkNode &nd : sk) maxDepth = std::max(maxDepth, nd.depth);
    std::vector<std::vector<int>> byDepth((size_t)maxDepth + 1);
    for (int i = 0; i < (int)sk.size(); ++i) byDepth[(size_t)sk[(size_t)i].depth].push_back(i);
    for (auto &v : byDepth) std::sort(v.begin(), v.end(), [&](int a, int b) { return sk[(size_t)a].begin < sk[(size_t)b].begin; });
    size_t maxSeg = 0;
    for (auto &v : byDepth) maxSeg = std::max(maxSeg, v.size());
    std::vector<int> outOfPos((size_t)nTris);
    for (const SkNode &nd : sk)
        if (nd.left < 0) for (int k = 0; k < nd.end - nd.begin; ++k) outOfPos[(size_t)(nd.begin + k)] = nd.firstOut + k;

    const int n = nTris;
    const unsigned gN = (unsigned)((n + 255) / 256);
    float *dT9 = nullptr, *dMn = nullptr, *dMx = nullptr, *dCen = nullptr, *dT12 = nullptr;
    int *dPerm[2] = {nullptr, nullptr}, *dSegB = nullptr, *dSegE = nullptr, *dSegI = nullptr, *dOut = nullptr;
    unsigned long long *dKeys[2] = {nullptr, nullptr};
    uint32_t *dBounds = nullptr;
    void *dTemp = nullptr;
    size_t tempBytes = 0;
    std::vector<uint32_t> hostBounds(sk.size() * 6);
    std::vector<int> segB, segE, segI;
    int cur = 0;
    (void)hipSetDevice(device);
    hipStream_t st = nullptr;
    BG_TRY(hipStreamCreate(&st));
    BG_TRY(hipMalloc(&dT9, (size_t)n * 9 * 4)); BG_TRY(hipMalloc(&dMn, (size_t)n * 3 * 4)); BG_TRY(hipMalloc(&dMx, (size_t)n * 3 * 4));
    BG_TRY(hipMalloc(&dCen, (size_t)n * 3 * 4)); BG_TRY(hipMalloc(&dT12, (size_t)n * 12 * 4));
    BG_TRY(hipMalloc(&dPerm[0], (size_t)n * 4)); BG_TRY(hipMalloc(&dPerm[1], (size_t)n * 4));
    BG_TRY(hipMalloc(&dKeys[0], (size_t)n * 8)); BG_TRY(hipMalloc(&dKeys[1], (size_t)n * 8));
    BG_TRY(hipMalloc(&dSegB, maxSeg * 4)); BG_TRY(hipMalloc(&dSegE, maxSeg * 4)); BG_TRY(hipMalloc(&dSegI, maxSeg * 4));
    BG_TRY(hipMalloc(&dBounds, maxSeg * 6 * 4)); BG_TRY(hipMalloc(&dOut, (size_t)n * 4));
    BG_TRY(hipMemcpyAsync(dT9, tris9, (size_t)n * 9 * 4, hipMemcpyHostToDevice, st));
    BG_TRY(hipMemcpyAsync(dOut, outOfPos.data(), (size_t)n * 4, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_tri_prep, dim3(gN), dim3(256), 0, st, dT9, n, dMn, dMx, dCen);
    hipLaunchKernelGGL(k_iota, dim3(gN), dim3(256), 0, st, dPerm[0], n);
    {
        rocprim::double_buffer<unsigned long long> kb(dKeys[0], dKeys[1]);
        rocprim::double_buffer<int> vb(dPerm[0], dPerm[1]);
        BG_TRY(rocprim::radix_sort_pairs(nullptr, tempBytes, kb, vb, (size_t)n, 0, 64, st));
        BG_TRY(hipMalloc(&dTemp, std::max<size_t>(tempBytes, 16)));
    }
    for (int d = 0; d <= maxDepth; ++d) {
        const std::vector<int> &ids = byDepth[(size_t)d];
        const int nSeg = (int)ids.size();
        segB.resize((size_t)nSeg); segE.resize((size_t)nSeg); segI.resize((size_t)nSeg);
        bool anyInner = false;
        for (int s = 0; s < nSeg; ++s) {
            const SkNode &nd = sk[(size_t)ids[(size_t)s]];
            segB[(size_t)s] = nd.begin; segE[(size_t)s] = nd.end; segI[(size_t)s] = nd.left >= 0;
            anyInner = anyInner || nd.left >= 0;
        }
        BG_TRY(hipMemcpyAsync(dSegB, segB.data(), (size_t)nSeg * 4, hipMemcpyHostToDevice, st));
        BG_TRY(hipMemcpyAsync(dSegE, segE.data(), (size_t)nSeg * 4, hipMemcpyHostToDevice, st));
        BG_TRY(hipMemcpyAsync(dSegI, segI.data(), (size_t)nSeg * 4, hipMemcpyHostToDevice, st));
        // identity of min over sortable uints = 0xffffffff, of max = 0: fill [min,min,min,max,max,max] per node
        {
            std::vector<uint32_t> init((size_t)nSeg * 6);
            for (int s = 0; s < nSeg; ++s) for (int c = 0; c < 6; ++c) init[(size_t)s * 6 + c] = c < 3 ? 0xffffffffu : 0u;
            BG_TRY(hipMemcpyAsync(dBounds, init.data(), init.size() * 4, hipMemcpyHostToDevice, st));
            BG_TRY(hipStreamSynchronize(st));   // `init` and the seg vectors are reused next level
        }
        hipLaunchKernelGGL(k_level_bounds, dim3(gN), dim3(256), 0, st, dPerm[cur], n, dMn, dMx, dSegB, dSegE, nSeg, dBounds);
        {
            std::vector<uint32_t> got((size_t)nSeg * 6);
            BG_TRY(hipMemcpyAsync(got.data(), dBounds, got.size() * 4, hipMemcpyDeviceToHost, st));
            BG_TRY(hipStreamSynchronize(st));
            for (int s = 0; s < nSeg; ++s) std::memcpy(&hostBounds[(size_t)ids[(size_t)s] * 6], &got[(size_t)s * 6], 24);
        }
        if (!anyInner) continue;
        hipLaunchKernelGGL(k_level_keys, dim3(gN), dim3(256), 0, st, dPerm[cur], n, dCen, dSegB, dSegE, dSegI, nSeg, dBounds, dKeys[cur]);
        {
            rocprim::double_buffer<unsigned long long> kb(dKeys[cur], dKeys[cur ^ 1]);
            rocprim::double_buffer<int> vb(dPerm[cur], dPerm[cur ^ 1]);
            BG_TRY(rocprim::radix_sort_pairs(dTemp, tempBytes, kb, vb, (size_t)n, 0, 64, st));
            cur = (vb.current() == dPerm[0]) ? 0 : 1;
        }
    }
    hipLaunchKernelGGL(k_emit_tris, dim3(gN), dim3(256), 0, st, dT9, dPerm[cur], dOut, n, dT12);
    BG_TRY(hipMemcpyAsync(tris12, dT12, (size_t)n * 12 * 4, hipMemcpyDeviceToHost, st));
    BG_TRY(hipStreamSynchronize(st));
    for (size_t i = 0; i < sk.size(); ++i) {   // upload_bvh_tbo node texels, bvh.cpp:153-168
        const SkNode &nd = sk[i];
        float *o = nodes12 + i * 12;
        for (int c = 0; c < 6; ++c) {
            const uint32_t s = hostBounds[i * 6 + (size_t)c];
            const uint32_t u = (s & 0x80000000u) ? (s & 0x7fffffffu) : ~s;
            std::memcpy(&o[c < 3 ? c : c + 1], &u, 4);
        }
        o[3] = (float)nd.left; o[7] = (float)nd.right;
        o[8] = nd.left < 0 ? (float)nd.firstOut : -1.0f;
        o[9] = nd.left < 0 ? (float)(nd.end - nd.begin) : 0.0f;
        o[10] = o[11] = 0.0f;
    }
done:
    for (void *p : {(void *)dT9, (void *)dMn, (void *)dMx, (void *)dCen, (void *)dT12, (void *)dPerm[0], (void *)dPerm[1], (void *)dKeys[0], (void *)dKeys[1],
                    (void *)dSegB, (void *)dSegE, (void *)dSegI, (void *)dBounds, (void *)dOut, dTemp})
        if (p) (void)hipFree(p);
    if (st) (void)hipStreamDestroy(st);
    if (err) { if (errOut) *errOut = err; return RT_ERR_HIP; }
    return (int)sk.size();
}

}  // namespace rtl

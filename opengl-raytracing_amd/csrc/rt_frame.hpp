// rt_frame.hpp -- device-resident frame description shared by all kernels, the framebuffer tile
// layout, and the launcher prototypes that connect rt_api.hip to the kernel translation units.
//
// Framebuffer layout in HBM ("tile-major"): the W x H frame is cut into 16x16-pixel tiles,
// globalTile = tileY * tilesX + (tileX + rowShift(tileY)) % tilesX.  Rank r of a world of n owns the tiles with
// globalTile % n == r and stores them densely: localTile = globalTile / n.  rowShift (round 5) is 0 on one
// GPU and 11 * tileY mod tilesX on several: tilesX is 120 at 1080p and 240 at 4K -- a multiple of every world size from 2 to 8 but
// 7 -- so without it a rank owned fixed 16-pixel COLUMNS of the frame and anything with vertical structure at that period landed on
// one rank; with it a rank's tiles move 11 columns (coprime to every world size up to 10) from row to row.  A tile is 256
// consecutive pixels = one 256-thread workgroup; inside it each 64-pixel run is one 8x8 block =
// one wavefront, so a wave's loads/stores of a target are one contiguous 512-byte (RGBA16F) run
// and its rays start out spatially coherent.
//   pixel slot = localTile * 256 + q * 64 + lane,   q = 8x8 quadrant (bit0: x half, bit1: y half),
//   x = tileX*16 + (q&1)*8 + (lane&7),  y = tileY*16 + (q>>1)*8 + (lane>>3)      (y = 0 is the bottom row)
#pragma once
#include <hip/hip_runtime.h>

#include "rt_device_shade.hpp"

namespace rtd {

#define RT_MAX_BATCH 16
struct FrameGeom {
    int W, H, tilesX, tilesY, nTiles, rank, world, nLocalTiles;
    // Frame batching (rt_render_frames): `batch` consecutive frames of a static camera share one set of launches.  The kernels see
    // batch x nLocalTiles "local tiles": local tile index lt' = k * nLocalTiles + lt is tile lt of the batch's k-th frame, and pixel
    // slots follow (slot' = lt' * 256 + tid).  batch == 1 is the plain single frame.
    int batch;
};

struct DevFrame {   // one copy in HBM, refreshed per frame; kernels read it through scalar loads
    RtUniforms u;
    DevScene sc;
    FrameGeom g;
    int giBounces;  // EXTENSION (rt_set_extension): bounces of the analytic / hybrid GI path, 1 = the reference
    float jitterK[RT_MAX_BATCH][2];   // uJitter of the batch's frames (frame k has uFrameIndex = u.frameIndex + k); [0] == u.jitter
};

struct Targets {
    uint2 *color;        // COLOR0 write ping  (RGBA16F: rgb + luma second moment), rt.frag:29
    const uint2 *prev;   // COLOR0 read pong   (uPrevAccum), this rank's tiles
    const uint2 *prevAll;   // tile-parallel + moving camera: every rank's COLOR0 block of the previous frame, rank-major (or null)
    int blockSlots;         // slots per rank block in prevAll
    uint32_t *motion;    // COLOR1 RG16F
    uint2 *gpos;         // COLOR2 RGBA16F
    uint2 *gnrm;         // COLOR3 RGBA16F
};

RT_DEV int sub_frame_of_tile(const FrameGeom &g, int localTile) { return g.batch > 1 ? localTile / max(g.nLocalTiles, 1) : 0; }
RT_DEV int sub_frame_of_slot(const FrameGeom &g, uint32_t slot) { return sub_frame_of_tile(g, (int)(slot >> 8)); }
// the tile deal: global tile index of tile (tx, ty) and back.  Row ty's tiles are numbered from column rowShift on (see the header comment).
constexpr int kTileRowShift = 11;
__host__ __device__ inline int tile_row_shift(const FrameGeom &g, int ty) { return g.world > 1 ? (ty * kTileRowShift) % g.tilesX : 0; }
__host__ __device__ inline int tile_index(const FrameGeom &g, int tx, int ty) { return ty * g.tilesX + (tx + tile_row_shift(g, ty)) % g.tilesX; }
__host__ __device__ inline void tile_xy(const FrameGeom &g, int t, int &tx, int &ty) {
    ty = t / g.tilesX;
    tx = (t % g.tilesX + g.tilesX - tile_row_shift(g, ty)) % g.tilesX;
}
RT_DEV bool pixel_of_slot(const FrameGeom &g, int localTile, int tid, int &x, int &y) {
    if (g.batch > 1) {
        if (localTile >= g.nLocalTiles * g.batch) { x = y = 0; return false; }
        localTile %= max(g.nLocalTiles, 1);
    }
    int t = localTile * g.world + g.rank;
    if (t >= g.nTiles) { x = y = 0; return false; }   // padding of the last local tile row of this rank
    int tx, ty;
    tile_xy(g, t, tx, ty);
    int q = tid >> 6, lane = tid & 63;
    x = tx * 16 + (q & 1) * 8 + (lane & 7);
    y = ty * 16 + (q >> 1) * 8 + (lane >> 3);
    return t < g.nTiles && x < g.W && y < g.H;
}
// slot of pixel (x,y) if this rank owns it, else -1
RT_DEV int slot_of_pixel(const FrameGeom &g, int x, int y) {
    int tx = x >> 4, ty = y >> 4;
    int t = tile_index(g, tx, ty);
    if (t % g.world != g.rank) return -1;
    int lx = x & 15, ly = y & 15;
    int q = (lx >> 3) | ((ly >> 3) << 1);
    return (t / g.world) * 256 + q * 64 + (ly & 7) * 8 + (lx & 7);
}

RT_DEV uint2 pack_half4(V4 v) {
    uint2 r;
    r.x = (uint32_t)f32_to_f16_bits(v.x) | ((uint32_t)f32_to_f16_bits(v.y) << 16);
    r.y = (uint32_t)f32_to_f16_bits(v.z) | ((uint32_t)f32_to_f16_bits(v.w) << 16);
    return r;
}
RT_DEV uint32_t pack_half2(V2 v) { return (uint32_t)f32_to_f16_bits(v.x) | ((uint32_t)f32_to_f16_bits(v.y) << 16); }
RT_DEV V4 unpack_half4(uint2 r) {
    return mk4(f16_bits_to_f32((uint16_t)(r.x & 0xffffu)), f16_bits_to_f32((uint16_t)(r.x >> 16)),
               f16_bits_to_f32((uint16_t)(r.y & 0xffffu)), f16_bits_to_f32((uint16_t)(r.y >> 16)));
}

// slot of pixel (x,y) in a rank-major array of gathered blocks (blockSlots slots per rank), whoever owns it
RT_DEV int slot_in_gathered(const FrameGeom &g, int x, int y, int blockSlots) {
    int tx = x >> 4, ty = y >> 4;
    int t = tile_index(g, tx, ty);
    int lx = x & 15, ly = y & 15;
    int q = (lx >> 3) | ((ly >> 3) << 1);
    return (t % g.world) * blockSlots + (t / g.world) * 256 + q * 64 + (ly & 7) * 8 + (lx & 7);
}

// History access for resolveTAA (rt_taa.glsl:87,128): NEAREST, CLAMP_TO_EDGE.
struct HistoryTex {
    const uint2 *prev;
    const uint2 *prevAll;
    int blockSlots;
    const FrameGeom *g;
    int slot;
    RT_DEV V4 own() const { return unpack_half4(prev[slot]); }
    RT_DEV V4 at(float u, float v) const {
        int x = (int)__builtin_floorf(u * (float)g->W), y = (int)__builtin_floorf(v * (float)g->H);
        x = min(max(x, 0), g->W - 1);
        y = min(max(y, 0), g->H - 1);
        if (prevAll) return unpack_half4(prevAll[slot_in_gathered(*g, x, y, blockSlots)]);   // reprojection crosses tiles: exchanged history
        int s = slot_of_pixel(*g, x, y);
        return unpack_half4(prev[s < 0 ? slot : s]);   // s < 0 only with world > 1, where the host insists on the exchanged history
    }
};

// Block-wide accumulation of per-lane work counters into the 10 global 64-bit counters.
RT_DEV void flush_work(const Work &w, unsigned long long *counters) {
    const uint32_t v[10] = {w.raysClosest, w.raysShadow, w.raysAnalytic, w.nodeFetch, w.triFetch, w.envLookup, w.hitPixels,
                            w.fetchPrimary, w.fetchShadow, w.fetchAO};
    for (int i = 0; i < 10; ++i) {
        unsigned long long s = v[i];
        for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
        if ((threadIdx.x & 63) == 0 && s) atomicAdd(&counters[i], s);
    }
}

}  // namespace rtd

// ---- launchers implemented in the kernel translation units --------------------------------------
struct RtWaveBuffers;   // rt_wave.hip
namespace rtl {
hipError_t launch_present(hipStream_t s, const rtd::FrameGeom &g, const uint2 *color, const uint32_t *motion, const uint2 *gpos,
                          const uint2 *gnrm, const RtPresentParams &p, uint32_t *outRGBA8, int gatheredBlockSlots = 0);
// rt_bvh_gpu.hip: the reference's median-split builder on the device -> number of nodes or a negative RtStatus
int build_bvh_gpu(int device, const float *tris9, int nTris, float *nodes12, float *tris12, const char **errOut);
hipError_t launch_mega(hipStream_t s, const rtd::DevFrame *frame, rtd::Targets tg, unsigned long long *counters, bool count,
                       int stackDepth, int nLocalTiles);
}

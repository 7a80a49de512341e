/* include/rt_mi355.h -- C ABI of librt_mi355.so, the MI355X (gfx950) drop-in for the reference's
 * ray-trace pass.
 *
 * The reference (Darky-The-Dragon/OpenGL-RayTracing) has no plugin/FFI seam; the narrowest one is
 *     void renderRay(AppState&, int fbw, int fbh, bool cameraMoved,
 *                    const glm::mat4& currView, const glm::mat4& currProj);     include/render/render.h:19
 * whose real contract is "set ~75 uniforms, bind 4 resources, draw one full-screen triangle"
 * (src/render/render.cpp:55-194).  Each entry point below names the reference interface it replaces.
 * INTEGRATION.md shows the call-site a maintainer would change.
 *
 * Conventions: plain pointers and sizes, no C++/torch types; every function returns RT_OK (0) or a
 * negative RtStatus and never throws; the caller owns all host pointers and the library copies
 * during the call; one context is driven from one thread at a time; the library owns device
 * memory and its HIP stream.  Matrices are column-major float[16] (glm / glUniformMatrix4fv with
 * transpose = GL_FALSE, src/render/Shader.cpp:190-192).  Images are row-major with ROW 0 = BOTTOM
 * row (GL window origin, what glReadPixels returns).
 */
#ifndef RT_MI355_H
#define RT_MI355_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum RtStatus {
    RT_OK = 0,
    RT_ERR_INVALID = -1,      /* bad argument */
    RT_ERR_NO_DEVICE = -2,    /* no HIP device / HIP runtime unusable: the product path never falls back to a CPU */
    RT_ERR_HIP = -3,          /* a HIP call failed; rt_last_error() has the text */
    RT_ERR_STATE = -4,        /* call order (e.g. render before resize) */
    RT_ERR_UNSUPPORTED = -5,  /* valid request this build cannot serve (stated in the message) */
    RT_ERR_IO = -6            /* file could not be read / parsed */
} RtStatus;

/* The uniform block of the ray-trace program: shaders/rt/rt_uniforms.glsl:25-177, same order.
 * All members are 4 bytes wide, so the struct has no padding. */
typedef struct RtUniforms {
    float eps, pi, inf;                                   /* uEPS uPI uINF  (RenderParams.h:229-231) */
    float camPos[3], camRight[3], camUp[3], camFwd[3];
    float tanHalfFov, aspect;
    int32_t frameIndex;                                   /* set by the library from its accumulation state */
    int32_t spp;
    float resolution[2];
    float jitter[2];
    int32_t enableJitter;
    int32_t useBVH, nodeCount, triCount;
    int32_t showMotion;
    float prevViewProj[16], currViewProj[16];
    int32_t cameraMoved;
    float taaStillThresh, taaHardMovingThresh;
    float taaHistoryMinWeight, taaHistoryAvgWeight, taaHistoryMaxWeight, taaHistoryBoxSize;
    int32_t enableTAA;
    float giScaleAnalytic, giScaleBVH;
    int32_t enableGI, enableAO, aoSamples;
    float aoRadius, aoBias, aoMin;
    int32_t useEnvMap;
    float envIntensity;
    int32_t sunEnabled;
    float sunColor[3], sunIntensity, sunDir[3];
    int32_t skyEnabled;
    float skyColor[3], skyIntensity, skyUpDir[3];
    int32_t pointLightEnabled;
    float pointLightPos[3], pointLightColor[3], pointLightIntensity;
    float matAlbedoColor[3], matAlbedoSpecStrength, matAlbedoGloss;
    float matGlassAlbedo[3], matGlassIOR, matGlassDistortion;
    int32_t matGlassEnabled;
    float matMirrorAlbedo[3], matMirrorGloss;
    int32_t matMirrorEnabled;
} RtUniforms;

/* include/render/RenderParams.h:14-239, same order and defaults (rt_default_render_params). */
typedef struct RtRenderParams {
    int32_t sppPerFrame; float exposure;
    float matAlbedoColor[3], matAlbedoSpecStrength, matAlbedoGloss;
    int32_t matGlassEnabled; float matGlassColor[3], matGlassIOR, matGlassDistortion;
    int32_t matMirrorEnabled; float matMirrorColor[3], matMirrorGloss;
    int32_t enableJitter; float jitterStillScale, jitterMovingScale;
    int32_t enableGI; float giScaleAnalytic, giScaleBVH;
    int32_t enableEnvMap; float envMapIntensity;
    int32_t sunEnabled; float sunColor[3], sunIntensity, sunYaw, sunPitch;
    int32_t skyEnabled; float skyColor[3], skyIntensity, skyYaw, skyPitch;
    int32_t pointLightEnabled; float pointLightColor[3], pointLightIntensity, pointLightPos[3];
    int32_t pointLightOrbitEnabled; float pointLightOrbitRadius, pointLightOrbitSpeed, pointLightYaw, pointLightPitch;
    int32_t enableAO, aoSamples; float aoRadius, aoBias, aoMin;
    int32_t enableTAA; float taaStillThresh, taaHardMovingThresh, taaHistoryMinWeight, taaHistoryAvgWeight,
        taaHistoryMaxWeight, taaHistoryBoxSize;
    int32_t enableSVGF; float svgfVarMax, svgfKVar, svgfKColor, svgfKVarMotion, svgfKColorMotion, svgfStrength;
    float motionScale;
} RtRenderParams;

/* Camera state, include/io/Camera.h:21-109 (degrees). */
typedef struct RtCamera { float pos[3], yaw, pitch, fov, aspect; } RtCamera;

/* Work counters in the reference's units: a "ray" is one call of traceBVH /
 * traceBVHShadow / traceAnalyticCore; node/tri fetches are nodeFetch()/triFetch() calls of
 * shaders/rt/rt_bvh.glsl (48 B each in the reference layout). */
typedef struct RtCounters {
    uint64_t raysClosest, raysShadow, raysAnalytic, nodeFetch, triFetch, envLookup, hitPixels;
    /* nodeFetch + triFetch split by the kind of ray: primary rays (rt.frag:86), traceBVHShadow rays, computeAO's rays
     * (rt_lighting.glsl:721-757); what is left belongs to the bounce's closest-hit rays. */
    uint64_t fetchPrimary, fetchShadow, fetchAO;
} RtCounters;

typedef enum RtPipeline {
    RT_PIPELINE_AUTO = 0,       /* wavefront pipeline for BVH scenes, megakernel for the analytic scene, staged replay for the hybrid extension */
    RT_PIPELINE_MEGAKERNEL = 1, /* one thread per pixel, the whole fragment program in one kernel */
    RT_PIPELINE_WAVEFRONT = 2   /* staged: primary -> ray generation -> persistent traversal -> combine */
} RtPipeline;

typedef struct RtDeviceConfig {
    int32_t device;        /* HIP device ordinal */
    int32_t rank;          /* tile-parallel rank of this context, 0 <= rank < worldSize */
    int32_t worldSize;     /* number of GPUs sharing one frame (1 = whole frame here) */
    int32_t pipeline;      /* RtPipeline */
    int32_t countWork;     /* non-zero: kernels maintain RtCounters (slower) */
    int32_t reserved[3];
} RtDeviceConfig;

typedef struct RtContext RtContext;

enum { RT_TARGET_COLOR = 0, RT_TARGET_MOTION = 1, RT_TARGET_GPOS = 2, RT_TARGET_GNRM = 3 };  /* rt.frag:29-38 */
enum { RT_FORMAT_F16 = 0, RT_FORMAT_F32 = 1 };

#define RT_TILE_DIM 16               /* framebuffer tiles are RT_TILE_DIM x RT_TILE_DIM pixels */
#define RT_TILE_PIXELS 256

/* ---------------------------------------------------------------- device side */

/* Replaces: GL context + FBO/texture creation (Application::initGLResources, application.cpp:195-205). */
int rt_create(const RtDeviceConfig *cfg, RtContext **out);
void rt_destroy(RtContext *ctx);
const char *rt_last_error(const RtContext *ctx);   /* ctx may be NULL: error of the last failed rt_create */

/* Replaces upload_bvh_tbo (include/scene/bvh.h:121, src/scene/bvh.cpp:141-221): takes the reference's
 * two RGBA32F texture-buffer payloads (12 floats per node, 12 per triangle) and repacks them into
 * the device layout.  nNodes == 0 / nTris == 0 clears the scene. */
int rt_upload_bvh(RtContext *ctx, const float *nodes12, int nNodes, const float *tris12, int nTris);

/* Replaces the glTexImage2D face uploads of loadCubeMapFromCross / createDummyCubeMap
 * (src/render/cubemap.cpp:7-31, 67-91): 6 faces in GL order +X -X +Y -Y +Z -Z, faceSize^2 texels of
 * `channels` (3 or 4) bytes, rows in upload order.  faces == NULL installs the 1x1 dummy
 * (128,128,255) of cubemap.cpp:13. */
/* The same builder as rt_build_bvh (below), run on the context's GPU: identical node numbering, ranges and boxes and the same
 * set of triangles in every leaf whenever no two triangles tie at a median; the order of triangles INSIDE a leaf differs
 * (the reference's std::nth_element leaves it unspecified), so rt_build_bvh remains the bit-parity path and this the fast one.
 * Returns the number of nodes (or a negative RtStatus); nodes12 needs room for 2*nTris nodes. */
int rt_build_bvh_gpu(RtContext *ctx, const float *tris9, int nTris, float *nodes12, float *tris12);

int rt_upload_env(RtContext *ctx, const uint8_t *faces, int faceSize, int channels);

/* Replaces Accum::recreate + GBuffer::recreate (src/render/accum.cpp:106-, gbuffer.cpp:12-53):
 * allocates COLOR0 ping-pong (RGBA16F), motion (RG16F), world pos / normal (RGBA16F); clears
 * history; frameIndex = 0. */
int rt_resize(RtContext *ctx, int width, int height);

/* Replaces Accum::reset (src/render/accum.cpp:98-102). */
int rt_reset_accum(RtContext *ctx);

/* accum.frameIndex (include/render/accum.h:125-128): the value the next frame will see as uFrameIndex. */
int rt_frame_index(const RtContext *ctx);

/* Replaces the ray pass of renderRay (src/render/render.cpp:58-194 + swapAfterFrame :242; the present
 * pass :199-239 is not part of this path).  `u` is the uniform block; u->frameIndex is ignored and
 * replaced by rt_frame_index().  Asynchronous on the context's stream. */
int rt_render_frame(RtContext *ctx, const RtUniforms *u);

/* `count` consecutive frames (us[i] = the uniform block of frame rt_frame_index() + i) with as few launches as possible: runs of
 * frames that differ only in uJitter / uFrameIndex and have uCameraMoved == 0 -- an accumulating static camera, every BASELINE
 * configuration -- are rendered up to 16 at a time by one set of kernel launches (BVH scenes, wavefront pipeline); anything else
 * falls back to one rt_render_frame per frame.  In the reference this is `count` turns of Application::mainLoop with a standing camera
 * (src/app/application.cpp:381-459: beginFrame, cameraMoved == false, the jitter of :398-405, renderRay, endFrame).
 * Bit-identical to `count` calls of rt_render_frame; afterwards the four targets hold
 * the last frame.  This is what keeps a tile-parallel rank busy: with 1/8 of the pixels a single frame is too little work per launch. */
int rt_render_frames(RtContext *ctx, const RtUniforms *us, int count);

/* The reference call-site in one call: mainLoop steps application.cpp:381-405 + renderRay + endFrame
 * (:459).  Keeps FrameState (prev/curr view-projection) inside the context.  currView/currProj may
 * be NULL: they are then derived from `cam` (Camera.cpp:66-73). */
int rt_render_ray(RtContext *ctx, const RtRenderParams *params, const RtCamera *cam, int useBVH, int showMotion,
                  const float *currView, const float *currProj);

/* ---- EXTENSION, not in the reference (BASELINE configs[2-3] "bunny + glass + mirror ..., 4 bounces": the reference's
 * BVH mode has neither analytic objects nor materials, rt.frag:84-106).  RtUniforms.useBVH == RT_SCENE_HYBRID renders the ANALYTIC
 * branch of rt.frag (:108-163) with the uploaded BVH mesh added to the analytic scene as one more object (material id 5: the default
 * branch of getMaterial, rt_materials.glsl:123-124), so the glass sphere refracts it, the mirror reflects it, it casts and receives
 * shadows, AO and GI.  giBounces > 1 lengthens the analytic GI path (oneBounceGIAnalytic) to that many diffuse bounces.  With an empty
 * BVH and giBounces == 1 this is the reference's analytic mode bit for bit.  Parity: this repository's own oracle only.
 * Pipelines: staged (RT_PIPELINE_AUTO / _WAVEFRONT: shading passes that replay answered mesh queries and queue the open ones, persistent
 * closest-hit traversal launches in between; csrc/rt_hybrid.hip) or the megakernel (RT_PIPELINE_MEGAKERNEL); same frames bit for bit. */
#define RT_SCENE_HYBRID 2
/* envFilter: model of texture(uEnvMap, dir)'s LINEAR filter (src/render/cubemap.cpp:56-58 GL_RGB8, :95-102 LINEAR / CLAMP_TO_EDGE).
 *   0 (default): bilinear weights from the fractional texel coordinates in exact fp32.
 *   1: the texel coordinates u = s*N - 0.5, v = t*N - 0.5 are first rounded to nearest on a grid of 1/256 texel (8 fractional bits of
 *      sub-texel precision, as GPU samplers filter RGB8), the weights (1-a)(1-b) ... then follow exactly; kept so that a capture from
 *      a real GL driver can be compared under either model.  Same texels, same face selection, same clamping. */
typedef struct RtExtension { int32_t giBounces; int32_t envFilter; int32_t reserved[2]; } RtExtension;
int rt_set_extension(RtContext *ctx, const RtExtension *ext);   /* applies to the frames rendered after the call */

/* `count` rt_render_ray calls with an unchanged camera and unchanged parameters, rendered through rt_render_frames (batched). */
int rt_render_ray_frames(RtContext *ctx, const RtRenderParams *params, const RtCamera *cam, int useBVH, int showMotion, int count);

int rt_synchronize(RtContext *ctx);

/* Read one render target of the last frame into host memory: full width x height image, row 0 =
 * bottom, channels 4/2/4/4, as half bit patterns (RT_FORMAT_F16) or floats (RT_FORMAT_F32, exact
 * widening).  With worldSize > 1 only this rank's tiles are filled, the rest is zero. */
int rt_read_target(RtContext *ctx, int which, void *dst, int dstFormat);

/* Inverse of rt_read_target for RT_FORMAT_F16: overwrite one render target "of the last frame" from a full
 * width x height host image (row 0 = bottom); a rank takes its own tiles.  Writing RT_TARGET_COLOR replaces the
 * accumulation history the next frame reads (uPrevAccum) -- what glTexSubImage2D on Accum::readTex() would do in
 * the reference (src/render/accum.cpp:8-20) -- so an accumulation can be restored from a saved frame or a test can
 * hand the renderer a known history.  The frame index is not touched (see rt_render_frame: it comes from RtUniforms). */
int rt_write_target(RtContext *ctx, int which, const void *src, int srcFormat);

/* Present pass of renderRay (src/render/render.cpp:199-239 = shaders/rt/rt_present.frag): SVGF-lite 7x7 filter,
 * ACES, gamma 1/2.2 (or the motion visualisation) over the four targets of the last frame -> RGBA8, width*height*4
 * bytes, row 0 = bottom.  RtPresentParams = the uniforms of rt_present.frag:38-50; rt_make_present_params fills
 * them from RenderParams as render.cpp:209-235 does.  Single-rank contexts only (the 7x7 taps cross tile borders). */
typedef struct RtPresentParams {
    float exposure; int32_t showMotion; float motionScale; float resolution[2];
    float varMax, kVar, kColor, kVarMotion, kColorMotion, svgfStrength; int32_t enableSVGF;
} RtPresentParams;
void rt_make_present_params(const RtRenderParams *p, int showMotion, int fbw, int fbh, RtPresentParams *out);
int rt_present(RtContext *ctx, const RtPresentParams *p, uint8_t *dstRGBA8);

/* Tile-parallel plumbing for a host that runs the exchange itself (e.g. through torch.distributed on these device pointers;
 * the library's own RCCL path is rt_comm_init / rt_gather_frame / rt_exchange_history below).
 * Local layout: [localTile][RT_TILE_PIXELS][channels] halfs, localTile = globalTile / worldSize for
 * globalTile % worldSize == rank, globalTile = tileY * tilesX + (tileX + rowShift) % tilesX, rowShift = 0 for worldSize 1 and
 * (11 * tileY) % tilesX otherwise: a rank's tiles are scattered over the frame, not fixed columns (csrc/rt_frame.hpp, tiles.py). */
int rt_local_target(RtContext *ctx, int which, void **devPtr, size_t *bytes);
/* bytes every rank must contribute so that all ranks send equally sized blocks (padded local size) */
int rt_gather_block_bytes(const RtContext *ctx, int which, size_t *bytes);
/* On the gathering rank: gatheredDev holds worldSize blocks of rt_gather_block_bytes() in rank order;
 * writes the row-major full frame (halfs) to dstDev (device memory, width*height*channels*2 bytes). */
int rt_assemble_gathered(RtContext *ctx, int which, const void *gatheredDev, void *dstDev);
/* the context's HIP stream (hipStream_t) so a caller can order its own work after the frame */
int rt_stream(RtContext *ctx, void **hipStream);

/* Tile-parallel frame with a MOVING camera.  Reprojection (rt_taa.glsl:116-179) reads the previous frame at arbitrary
 * pixels, i.e. in other ranks' tiles, so every rank needs the whole previous COLOR0.  After rt_render_frame(f) the host
 * all-gathers the ranks' COLOR0 blocks (rt_local_target / rt_gather_block_bytes) into the buffer returned by
 * rt_history_exchange_buffer -- worldSize blocks, rank-major, on rt_stream() -- and then calls rt_history_exchanged();
 * frame f+1 may then be rendered with cameraMoved = 1 (without the exchange: RT_ERR_STATE).  Static-camera frames need
 * no exchange: a pixel only reads its own history.  Both calls refer to the frame rendered last. */
int rt_history_exchange_buffer(RtContext *ctx, void **devPtr, size_t *bytes);
int rt_history_exchanged(RtContext *ctx);

/* Present pass over a tile-parallel frame on the gathering rank: the four arguments are device arrays of worldSize gathered
 * blocks each (as filled by the gather of COLOR / MOTION / GPOS / GNRM, rank-major, rt_gather_block_bytes per block). */
int rt_present_gathered(RtContext *ctx, const RtPresentParams *p, const void *gatheredColor, const void *gatheredMotion,
                        const void *gatheredGPos, const void *gatheredGNrm, uint8_t *dstRGBA8);

/* ---- the exchange itself, owned by the library ("the library owns device memory, streams and the RCCL
 * communicator"; 8e: gather for a static camera, all-gather of the history for a moving one).  One process per GPU; every call
 * below is collective over the ranks of the frame (RtDeviceConfig.rank / worldSize) and asynchronous on the stream of the frame
 * rendered last.  A C++ host needs nothing else to render tile-parallel: csrc/rt_cli.cpp --ranks N.
 *   rank 0:  rt_comm_unique_id(id)  -> hand the 128 bytes to the other ranks by any means (file, pipe, MPI, torch store)
 *   all:     rt_comm_init(ctx, id)  -> ncclCommInitRank
 *   per gathered frame:  rt_render_frame / rt_render_ray; rt_gather_frame(ctx, RT_TARGET_COLOR)   [+ rt_exchange_history]
 *   rank 0:  rt_read_gathered(ctx, RT_TARGET_COLOR, halfs)   or rt_gathered_frame() for the device pointer
 * Static camera: a pixel only reads its own history (rt_taa.glsl:86-105), which stays rank-local, so intermediate frames need
 * not be gathered at all -- call rt_gather_frame every k-th frame (BASELINE configs[4]: once per 32 accumulated frames). */
#define RT_COMM_ID_BYTES 128
int rt_comm_unique_id(void *id, size_t bytes);                         /* any process; needs librccl */
int rt_comm_init(RtContext *ctx, const void *id, size_t bytes);
int rt_comm_destroy(RtContext *ctx);                                   /* also done by rt_destroy */
int rt_gather_frame(RtContext *ctx, int which);                        /* worldSize == 1: a device copy + un-tiling, no RCCL */
int rt_gathered_frame(RtContext *ctx, int which, void **devPtr, size_t *bytes);   /* rank 0: row-major width x height halfs, row 0 = bottom */
int rt_read_gathered(RtContext *ctx, int which, void *dstHalfs);       /* rank 0: synchronises, copies that frame to the host */
int rt_present_last_gathered(RtContext *ctx, const RtPresentParams *p, uint8_t *dstRGBA8);   /* rank 0, after rt_gather_frame of all 4 targets */
int rt_exchange_history(RtContext *ctx);                               /* all-gather of COLOR0 + rt_history_exchanged() */
/* What the communicator itself reports (ncclCommCount / ncclCommUserRank; -1 = no communicator) and what the gathers of this
 * context moved: a tile-parallel run's line can then say which exchange really ran (bench.py's config.gather).  Device time of
 * the gathers: stage "gather" of rt_get_stage_times.  bytesIn: bytes received by the gathering rank (others: bytes sent). */
typedef struct RtCommInfo { int32_t commWorld, commRank, rank, worldSize; uint64_t gathers, gatherBytes, historyExchanges; } RtCommInfo;
int rt_comm_info(RtContext *ctx, RtCommInfo *out);

int rt_get_counters(RtContext *ctx, RtCounters *out);   /* needs countWork; totals since rt_reset_counters */
int rt_reset_counters(RtContext *ctx);

/* What rt_upload_bvh made of the scene: the device arrays of DESIGN.md 3 (64-byte two-child records for closest-hit rays, 128-byte
 * four-child records for any-hit rays, 80-byte triangle-pair records, the reference's 48-byte triangles for normals) and their
 * sizes -- the bytes a traversal launch has to bring in at most once (bench.py's HBM roofline).  When the any-hit tree is larger than
 * 4 MB the any-hit launches walk its quantised form instead: bytesNodes4 is then 64 bytes per four-child record + 32
 * bytes of exact box per leaf. */
#define RT_SCENE_QNODES_REJECTED 1   /* quantised any-hit nodes were asked for (tree size or RT_QNODES) but could not be built: the exact nodes are walked */
#define RT_SCENE_IMPLICIT 4          /* RT_IMPLICIT=1 and every leaf sits at one depth: closest-hit rays walk 48-byte records without child references */
#define RT_SCENE_NOT_FUSED 2         /* RT_FUSED=1 but some inner box is not the union of its children's: closest-hit rays walk the 64-byte records, not the fused ones */
typedef struct RtSceneInfo {
    int32_t nNodes, nTris, nInner, treeDepth, nWide4, nPairs;
    uint64_t bytesNodes2, bytesNodes4, bytesPairs, bytesTris;
    int32_t nFused;   /* fused closest-hit records (128 B per even-level inner node; 0: not built), DESIGN.md 4.2 */
    int32_t flags;    /* RT_SCENE_* */
    int32_t implicitDepth;   /* depth every leaf sits at when RT_SCENE_IMPLICIT is set */
    int32_t reserved;
} RtSceneInfo;
int rt_get_scene_info(const RtContext *ctx, RtSceneInfo *out);

/* Device memory behind the context: the ray-queue arenas of the wavefront pipeline (RtArenaPool: shared by the frame lanes), the
 * per-lane frame arrays (candidate / hit lists, pre-resolve stash), the hybrid extension's arena, and the device's free / total bytes
 * (hipMemGetInfo) -- bench.py reports them, so that the footprint of the timed mode is part of its line. */
typedef struct RtMemoryInfo { uint64_t queueArenaBytes, frameArrayBytes, hybridArenaBytes, deviceFreeBytes, deviceTotalBytes; int32_t queueArenas, lanes; } RtMemoryInfo;
int rt_get_memory_info(RtContext *ctx, RtMemoryInfo *out);

/* Rays the wavefront pipeline actually traversed since the last reset (identical rays of the reference -- the SPP
 * copies of a primary ray, the per-sample copies of the AO rays -- are traced once; disk-light shadow rays whose
 * weight is exactly zero are not traced at all).  RtCounters keeps counting in the reference's units.
 * gatherLoads*: 16-byte per-lane gather loads (BVH node and triangle records) the three traversal launches issued -- the unit of
 * the L1 gather roofline they run against (one divergent 16-byte lane-load per clock and CU, tools/gather.hip).
 * mergedLoads*: the same loads after merging the lanes of a wave that stand on the same record (they read the same 16-byte pieces,
 * which the vector L1 serves as ONE cache access): what the one-access-per-clock ceiling applies to.  Counted only by the
 * diagnostic traversal kernels (environment RT_TRACE_STATS=1 when the context renders); 0 otherwise. */
typedef struct RtTracedRays {
    uint64_t candidatePixels, hitPixels, primary, shadow, bounce, bounceShadow, frames;
    uint64_t gatherLoadsPrimary, gatherLoadsShadow, gatherLoadsBounce;
    uint64_t mergedLoadsPrimary, mergedLoadsShadow, mergedLoadsBounce;
    uint64_t ao, gatherLoadsAO;   /* AO rays traced as packets (one walk of the tree for the rays of a hit, round 4; not contained in `shadow`) and
                                   * the gather loads of that launch */
} RtTracedRays;
int rt_get_traced_rays(RtContext *ctx, RtTracedRays *out, int reset);

/* Device timing of the dominant kernel(s): HIP events recorded on the context's stream around each
 * stage of every frame since the last reset.  stage names: rt_stage_name(i). */
#define RT_MAX_STAGES 14
typedef struct RtStageTimes { int32_t nStages; int32_t frames; double ms[RT_MAX_STAGES]; uint64_t launches[RT_MAX_STAGES]; } RtStageTimes;
int rt_enable_stage_timing(RtContext *ctx, int enable);
int rt_get_stage_times(RtContext *ctx, RtStageTimes *out);   /* synchronises */
const char *rt_stage_name(int stage);

/* Diagnostics used by the parity tests: evaluate one device function of the float model on arrays
 * (op: 0 sin, 1 cos, 2 exp2, 3 log2, 4 pow(a,b), 5 f32->f16 bits, 6 rand(a,b,frame=c) bits,
 * 7 a/b, 8 sqrt(a), 9 1/sqrt(a)).  Arrays are host memory of n floats (out: n uint32 bit patterns). */
int rt_debug_eval(RtContext *ctx, int op, const float *a, const float *b, const float *c, uint32_t *out, int n);
/* Trace n rays against the uploaded BVH with the device traversal: kind 0 = closest hit (out: t, then
 * hit point xyz, then normal xyz; t = inf on miss), kind 1 = any hit within tMax (out[0] = 1/0).
 * kinds 2 / 3: the same two questions put to the wavefront pipeline's own traversal kernels (persistent launch, refill scheduler, the
 * any-hit node form rt_upload_bvh chose) -- kind 2: out[0] = t (inf on miss), out[1] = index of the triangle hit; kind 3: out[0] = 1/0. */
int rt_debug_trace(RtContext *ctx, int kind, const float *origins, const float *dirs, const float *tMax, float eps,
                   float inf, float *out7, int n);

/* ---------------------------------------------------------------- host side (no GPU needed) */

void rt_default_render_params(RtRenderParams *p);      /* include/render/RenderParams.h:20-238 */
void rt_default_camera(RtCamera *c);                   /* include/app/state.h:129-131 */
void rt_default_bvh_transform(float *M16);             /* include/app/state.h:26-31 */
void rt_camera_view(const RtCamera *c, float *V16);    /* Camera::GetViewMatrix, src/io/Camera.cpp:66-68 */
void rt_camera_proj(const RtCamera *c, float *P16);    /* Camera::GetProjectionMatrix, :71-73 */
void rt_mat4_mul(const float *A16, const float *B16, float *out16);   /* FrameState::beginFrame P*V, frame_state.h:71 */
void rt_generate_jitter(int frameIndex, float *out2);  /* generateJitter2D, src/app/application.cpp:42-47 */
int rt_camera_moved(const float *currVP16, const float *prevVP16);    /* application.cpp:387-395 */

/* The glUniform* block of renderRay, src/render/render.cpp:67-167, with the jitter policy of
 * application.cpp:398-405.  envLoaded = (app.envMapTex != 0). */
void rt_make_uniforms(const RtRenderParams *p, const RtCamera *cam, const float *currView, const float *currViewProj,
                      const float *prevViewProj, int fbw, int fbh, int frameIndex, int cameraMoved, int useBVH,
                      int showMotion, int nodeCount, int triCount, int envLoaded, RtUniforms *out);

/* gather_model_triangles (include/scene/bvh.h:135, src/scene/bvh.cpp:225-246): 9 floats (v0,e1,e2) per
 * index triple after the model matrix.  Returns the triangle count. */
int rt_gather_triangles(const float *positions, const uint32_t *indices, int nIdx, const float *M16, float *outTris9);
/* the same with the vertex count: RT_ERR_INVALID if any index is out of range (use this one for data read from files) */
int rt_gather_triangles_checked(const float *positions, int nVerts, const uint32_t *indices, int nIdx, const float *M16, float *outTris9);

/* build_bvh (include/scene/bvh.h:102, src/scene/bvh.cpp:94-137) + the packing half of upload_bvh_tbo
 * (:147-204).  nodes12 needs room for 2*nTris*12 floats, tris12 for nTris*12.  Returns the node count. */
int rt_build_bvh(const float *tris9, int nTris, float *nodes12, float *tris12);

/* Stand-in for Model/Mesh + Assimp (include/scene/model.h:105-228) for plain .obj files: v / f records,
 * fan triangulation, negative indices.  Buffers are malloc'ed; release with rt_free. */
int rt_load_obj(const char *path, float **positions, int *nVerts, uint32_t **indices, int *nIdx);
/* PNG decode (8-bit RGB / RGBA / grey, non-interlaced; zlib) standing in for stbi_load at cubemap.cpp:40 */
int rt_load_png(const char *path, uint8_t **pixels, int *width, int *height, int *channels);
/* 8-bit PNG writer (zlib), rows top-to-bottom as stored; flipY != 0 writes the last row first, i.e. turns a
 * bottom-up GL image the right way round */
int rt_save_png(const char *path, const uint8_t *pixels, int width, int height, int channels, int flipY);
void rt_free(void *p);

/* The 4x3 cross slicing of loadCubeMapFromCross (src/render/cubemap.cpp:47-91).  faces needs
 * 6*(height/3)^2*channels bytes.  Returns faceSize, 0 if the image is not a valid cross. */
int rt_cubemap_from_cross(const uint8_t *img, int width, int height, int channels, uint8_t *faces);

int rt_sizeof_uniforms(void);
int rt_sizeof_render_params(void);
const char *rt_version(void);

#ifdef __cplusplus
}
#endif
#endif /* RT_MI355_H */

// oracle/orc_types.h -- TEST INFRASTRUCTURE ONLY (parity oracle). PODs of the oracle's C ABI.
// Field order of OrcUniforms follows shaders/rt/rt_uniforms.glsl:25-177 so that one ctypes struct
// in tests/ can feed the oracle and the product (include/rt_mi355.h RtUniforms) the same bytes.
#pragma once
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct OrcUniforms {
    float eps, pi, inf;                                   // rt_uniforms.glsl:25-27  (RenderParams.h:229-231)
    float camPos[3], camRight[3], camUp[3], camFwd[3];    // :30-33
    float tanHalfFov, aspect;                             // :34-35
    int32_t frameIndex, spp;                              // :38-39
    float resolution[2];                                  // :40
    float jitter[2];                                      // :46
    int32_t enableJitter;                                 // :47
    int32_t useBVH, nodeCount, triCount;                  // :50-52
    int32_t showMotion;                                   // :57
    float prevViewProj[16], currViewProj[16];             // :58-59, column-major (Shader.cpp:190-192)
    int32_t cameraMoved;                                  // :60
    float taaStillThresh, taaHardMovingThresh;            // :63-64
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
} OrcUniforms;

// include/render/RenderParams.h:14-239, same order, same defaults (orc_default_render_params).
typedef struct OrcRenderParams {
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
} OrcRenderParams;

// include/io/Camera.h:21-109 (state only).
typedef struct OrcCamera { float pos[3], yaw, pitch, fov, aspect; } OrcCamera;

// Uniforms of shaders/rt/rt_present.frag:38-50 as set at src/render/render.cpp:209-235.
typedef struct OrcPresentParams {
    float exposure; int32_t showMotion; float motionScale; float resolution[2];
    float varMax, kVar, kColor, kVarMotion, kColorMotion, svgfStrength; int32_t enableSVGF;
} OrcPresentParams;

typedef struct OrcCounters {
    uint64_t raysClosest;     // traceBVH calls
    uint64_t raysShadow;      // traceBVHShadow calls
    uint64_t raysAnalytic;    // traceAnalyticCore calls
    uint64_t nodeFetch;       // nodeFetch() calls, 48 B each (rt_bvh.glsl:91)
    uint64_t triFetch;        // triFetch() calls, 48 B each (rt_bvh.glsl:55)
    uint64_t envLookup;       // texture(uEnvMap, .) calls
    uint64_t hitPixels;       // pixels whose primary ray hit (s == 0)
    uint64_t fetchPrimary;    // node+tri fetches made by the primary rays (rt.frag:86)
    uint64_t fetchShadow;     //   ... by traceBVHShadow (direct and bounce shadow rays)
    uint64_t fetchAO;         //   ... by computeAO's closest-hit rays (rt_lighting.glsl:721-757); the rest: the bounce rays
} OrcCounters;

#ifdef __cplusplus
}
#endif

// rt_device_analytic.hpp -- the analytic test scene of the reference on the device:
// shaders/rt/rt_materials.glsl, rt_scene_analytic.glsl and the analytic branches of rt_lighting.glsl
// (directLight, oneBounceGIAnalytic, shadeGlass, shadeMirror, computeAO).  Pure ALU: five implicit
// primitives, no memory traffic besides the cube map, so it stays a one-thread-per-pixel kernel.
#pragma once
#include "rt_device_shade.hpp"

#pragma clang fp contract(off)

namespace rtd {

enum { MAT_FLOOR = 0, MAT_ALBEDO_SPHERE = 1, MAT_GLASS_SPHERE = 2, MAT_MIRROR_SPHERE = 3, MAT_POINTLIGHT_SPHERE = 4,
       MAT_MESH = 5 };   // EXTENSION (hybrid scene): no material of rt_materials.glsl -> getMaterial's default branch (:123-124)

RT_DEV MaterialProps mkMat(V3 albedo, float spec, float gloss, int type, float ior) {
    MaterialProps m;
    m.albedo = albedo; m.specStrength = spec; m.gloss = gloss; m.type = type; m.ior = ior;
    return m;
}
RT_DEV MaterialProps getMaterial(const RtUniforms &u, int id) {   // rt_materials.glsl:57-125
    MaterialProps alb = mkMat(ld3(u.matAlbedoColor), u.matAlbedoSpecStrength, u.matAlbedoGloss, 0, 1.0f);
    if (id == MAT_FLOOR) return mkMat(mk3(0.7f), 0.1f, 16.0f, 0, 1.0f);
    if (id == MAT_ALBEDO_SPHERE) return alb;
    if (id == MAT_GLASS_SPHERE) {
        if (u.matGlassEnabled == 0) return alb;
        return mkMat(ld3(u.matGlassAlbedo), u.matGlassDistortion, 1.0f, 2, u.matGlassIOR);
    }
    if (id == MAT_MIRROR_SPHERE) {
        if (u.matMirrorEnabled == 0) return alb;
        return mkMat(ld3(u.matMirrorAlbedo), 0.0f, u.matMirrorGloss, 1, 1.0f);
    }
    return mkMat(mk3(0.8f), 0.2f, 16.0f, 0, 1.0f);
}

RT_DEV bool intersectPlane(float eps, V3 ro, V3 rd, V3 n, float d, Hit &h, int matId) {   // rt_scene_analytic.glsl:71-81
    float denom = dot(n, rd);
    if (__builtin_fabsf(denom) < 1e-6f) return false;
    float t = -(dot(n, ro) + d) / denom;
    if (t < eps) return false;
    h.t = t;
    h.p = ro + rd * t;
    h.n = n;
    h.mat = matId;
    return true;
}
RT_DEV bool intersectSphere(float eps, V3 ro, V3 rd, V3 c, float r, Hit &h, int matId) {   // :96-111
    V3 oc = ro - c;
    float b = dot(oc, rd);
    float c2 = dot(oc, oc) - r * r;
    float disc = b * b - c2;
    if (disc < 0.0f) return false;
    float s = __builtin_sqrtf(disc);
    float t = -b - s;
    if (t < eps) t = -b + s;
    if (t < eps) return false;
    h.t = t;
    h.p = ro + rd * t;
    h.n = normalize(h.p - c);
    h.mat = matId;
    return true;
}
// Inlining policy of the analytic scene's four big functions.  The megakernel keeps them as calls (its analytic branch is cold next to the BVH branch); a kernel
// that does nothing else may define RT_ANALYTIC_LEAF / RT_ANALYTIC_MID before including this header (rt_hybrid.hip, round 5): as calls, the Replay state, every Hit
// and every out-parameter live in scratch memory and each call saves and restores registers there (profiles/r05_hybrid_pmc.txt).
#ifndef RT_ANALYTIC_LEAF
#define RT_ANALYTIC_LEAF __noinline__      // traceAnalyticCore, traceScene: ~40 call sites per sample
#endif
#ifndef RT_ANALYTIC_MID
#define RT_ANALYTIC_MID __noinline__       // directLightA, oneBounceGIAnalytic
#endif
template <bool COUNT>
__device__ RT_ANALYTIC_LEAF bool traceAnalyticCore(const RtUniforms &u, V3 ro, V3 rd, bool includeGlass, bool includeMarker, Hit &hit,
                                               Work &w) {   // :132-167
    if (COUNT) w.raysAnalytic++;
    hit.t = u.inf;
    Hit h;
    if (intersectPlane(u.eps, ro, rd, mk3(0.0f, 1.0f, 0.0f), 0.0f, h, MAT_FLOOR) && h.t < hit.t) hit = h;
    if (intersectSphere(u.eps, ro, rd, mk3(-1.2f, 1.0f, -3.5f), 1.0f, h, MAT_ALBEDO_SPHERE) && h.t < hit.t) hit = h;
    if (includeGlass) {
        if (intersectSphere(u.eps, ro, rd, mk3(0.7f, 1.0f, -5.0f), 1.0f, h, MAT_GLASS_SPHERE) && h.t < hit.t) hit = h;
    }
    if (intersectSphere(u.eps, ro, rd, mk3(1.2f, 0.7f, -2.5f), 0.7f, h, MAT_MIRROR_SPHERE) && h.t < hit.t) hit = h;
    if (includeMarker && u.pointLightEnabled == 1) {
        if (intersectSphere(u.eps, ro, rd, ld3(u.pointLightPos), 0.15f, h, MAT_POINTLIGHT_SPHERE) && h.t < hit.t) hit = h;
    }
    return hit.t < u.inf;
}

// Scene query of the analytic shading code.  Reference modes: traceAnalyticCore.  EXTENSION uUseBVH == RT_SCENE_HYBRID (SURVEY.md 8d
// config 3 "run B"): the BVH mesh is one more object at the end of the list, under the list's own rule (strict <: an earlier object
// wins a tie); its hit is what traceBVH returns (geometric normal), its material id MAT_MESH.
// `geometric`: the caller builds later rays from this hit (primary, bounce, reflection, refraction); false for visibility / AO queries,
// whose answers only scale radiance.  The staged pipeline speculates "miss" for every open query alike; when a speculation fails, the
// answers recorded behind a failed GEOMETRIC query are void (their rays were built from the wrong hit), those behind a visibility query are not.
template <bool COUNT>
__device__ RT_ANALYTIC_LEAF bool traceScene(const Frag &F, V3 ro, V3 rd, bool includeGlass, bool includeMarker, Hit &hit, Work &w, bool geometric = false) {
    const RtUniforms &u = *F.u;
    bool any = traceAnalyticCore<COUNT>(u, ro, rd, includeGlass, includeMarker, hit, w);
    if (u.useBVH == RT_SCENE_HYBRID && F.rp) {
        // Staged: the mesh query is answered from the log of earlier passes, or queued.  A ray that misses the mesh's root box is a miss at
        // once -- the same test bvh_closest starts with -- and takes no query number, in every pass alike.
        Replay &R = *F.rp;
        const DevScene &sc = *F.sc;
        float tmin;
        const V3 rdInv = mk3(1.0f / rd.x, 1.0f / rd.y, 1.0f / rd.z);
        if (sc.hasBVH && slab(ro, rdInv, ld3(sc.rootMin), ld3(sc.rootMax), tmin) && !(tmin > u.inf)) {
            const uint32_t q = R.q++;
            if (q < R.known) {
                const int tri = R.logTri[q];
                const float t = R.logT[q];
                if (tri >= 0 && t < hit.t) {
                    hit.t = t;
                    hit.p = ro + rd * t;
                    hit.n = tri_normal(sc, tri);
                    hit.mat = MAT_MESH;
                    any = true;
                }
            } else {
                // open: recorded, and answered "no mesh hit" -- a speculation the next pass checks against the traced answer (rt_hybrid.hip).  Everything
                // the thread does after it is right exactly if that answer, and every later speculated one, turns out to be a miss.
                R.pending++;
                if (q < R.qmax) {
                    const size_t a = (size_t)q * 256u + R.slot;
                    R.o[a] = make_float4(ro.x, ro.y, ro.z, hit.t);   // .w: the analytic scene's hit distance (uINF: none) -- a mesh hit behind it changes nothing
                    R.d[a] = make_float4(rd.x, rd.y, rd.z, geometric ? 1.0f : 0.0f);   // .w: later rays are built from this hit
                    R.recEnd = q + 1u;
                } else R.overflow = true;
            }
        }
        return any;
    }
    if (u.useBVH == RT_SCENE_HYBRID && F.stk) {
        float t;
        int tri;
        if (bvh_closest<COUNT>(*F.sc, ro, rd, u.eps, u.inf, F.stk, t, tri, w) && t < hit.t) {
            hit.t = t;
            hit.p = ro + rd * t;
            hit.n = tri_normal(*F.sc, tri);
            hit.mat = MAT_MESH;
            any = true;
        }
    }
    return any;
}

// directLight, rt_lighting.glsl:313-395 (analytic occlusion branch of occludedToward :55-58,
// sunDirect :133-135, pointDirect :203-205).
template <bool COUNT>
__device__ RT_ANALYTIC_MID V3 directLightA(const Frag &F, const Hit &h, int frame, V3 Vdir, Work &w) {
    const RtUniforms &u = *F.u;
    V3 N = normalize(h.n);
    MaterialProps mat = getMaterial(u, h.mat);
    V3 V = normalize(Vdir);
    if (mat.type == 1) {
        V3 R = reflect(-V, N);
        V3 col = (u.useEnvMap == 1) ? texture_cube<COUNT>(*F.sc, R, w) * u.envIntensity : sky<COUNT>(F, R, w);
        return col * mat.albedo;
    }
    if (mat.type == 2) {
        V3 R = reflect(-V, N);
        V3 refl = (u.useEnvMap == 1) ? texture_cube<COUNT>(*F.sc, R, w) * u.envIntensity : sky<COUNT>(F, R, w);
        V3 skyDiff = skyDirect(u, h.n, mat);
        return refl * mat.albedo + skyDiff;
    }
    V3 lt, lb;
    lightFrame(lt, lb);
    V2 rot = cpOffset(F.fcx, F.fcy, F.frameIndex);
    V3 sum = mk3(0.0f);
    for (int i = 0; i < 4; ++i) {
        DiskSample s = diskSample(F, h.p, N, frame, i, rot, lt, lb);
        Hit hh;
        bool occ = traceScene<COUNT>(F, s.ro, s.rd, true, true, hh, w) && hh.t < s.tMax;
        float vis = occ ? 0.0f : 1.0f;
        V3 Li = mk3(18.0f) * s.geom * vis;
        sum = sum + shadeLambertPhong(u.pi, N, V, s.L, Li, mat.albedo, mat.specStrength, mat.gloss);
    }
    sum = sum / 4.0f;
    // sunDirect
    V3 sun = mk3(0.0f);
    if (u.sunEnabled != 0) {
        V3 Vs = normalize(V);   // sunDirect re-normalises the already normalised V (rt_lighting.glsl:119)
        V3 L = normalize(-ld3(u.sunDir));
        float ndl = fmaxr(dot(N, L), 0.0f);
        if (ndl > 0.0f) {
            float e = epsForDist(1000.0f);
            V3 origin = h.p + N * e;
            Hit tmp;
            bool blocked = traceScene<COUNT>(F, origin, L, true, true, tmp, w);
            if (!blocked) {
                float specStrength = (mat.type == 0) ? mat.specStrength : 0.0f;
                sun = shadeLambertPhong(u.pi, N, Vs, L, ld3(u.sunColor) * u.sunIntensity, mat.albedo, specStrength, mat.gloss);
            }
        }
    }
    sum = sum + sun;
    sum = sum + skyDirect(u, h.n, mat);
    // pointDirect
    V3 pt = mk3(0.0f);
    if (u.pointLightEnabled != 0) {
        V3 Vp = normalize(V);   // pointDirect does the same (rt_lighting.glsl:186)
        V3 toL = ld3(u.pointLightPos) - h.p;
        float dist2 = dot(toL, toL);
        if (dist2 > 1e-6f) {
            float dist = __builtin_sqrtf(dist2);
            V3 L = toL / dist;
            float ndl = fmaxr(dot(N, L), 0.0f);
            if (ndl > 0.0f) {
                float e = epsForDist(dist);
                V3 origin = h.p + L * e;
                Hit tmp;
                bool blocked = traceScene<COUNT>(F, origin, L, true, false, tmp, w) && tmp.t < dist - e;
                if (!blocked) {
                    V3 Li = ld3(u.pointLightColor) * (u.pointLightIntensity / fmaxr(dist2, 1e-4f));
                    float specStrength = (mat.type == 0) ? mat.specStrength : 0.0f;
                    pt = shadeLambertPhong(u.pi, N, Vp, L, Li, mat.albedo, specStrength, mat.gloss);
                }
            }
        }
    }
    sum = sum + pt;
    return sum;
}

// oneBounceGIAnalytic, rt_lighting.glsl:473-507, generalised to F.giBounces diffuse bounces (EXTENSION; giBounces == 1 -- the default
// and all the reference does -- is that function operation for operation: 1 * x and 0 + x are exact).  Level k starts at hit h_k with
// seed_k (seed_{k+1} = seed_k * 131 + 17, shadeMirror's derivation for its nested GI, :692) and throughput T_k (T_0 = 1):
//   F = albedo(h_k) * (cos / pi);  Li = directLight(h_{k+1}) if the bounce ray hits, else sky(wi) and the path ends;
//   result += (T_k * F) * Li;   T_{k+1} = (T_k * F) * giScaleAnalytic.           Same operations, same order as the parity checker's CPU restatement.
constexpr int kMaxGiBounces = 8;
template <bool COUNT>
__device__ RT_ANALYTIC_MID V3 oneBounceGIAnalytic(const Frag &F, const Hit &h0, int frame, int seed, Work &w) {
    const RtUniforms &u = *F.u;
    V3 result = mk3(0.0f), T = mk3(1.0f);
    Hit h = h0;
    const int maxB = min(max(F.giBounces, 1), kMaxGiBounces);
    for (int k = 0; k < maxB; ++k) {
        MaterialProps mat0 = getMaterial(u, h.mat);
        V3 N0 = normalize(h.n);
        float o13 = (float)(int)((uint32_t)seed * 13u), o37 = (float)(int)((uint32_t)seed * 37u);
        V2 uu = mk2(randr(F.fcx + o13, F.fcy + o13, frame), randr(F.fcy + o37, F.fcx + o37, frame));
        V3 wi = sampleHemisphereCosine(u.pi, N0, uu);
        float cosTheta = fmaxr(dot(N0, wi), 0.0f);
        if (cosTheta <= 0.0f) break;
        V3 origin = h.p + N0 * u.eps;
        Hit h1;
        bool hit1 = traceScene<COUNT>(F, origin, wi, true, true, h1, w, true);
        V3 Li = hit1 ? directLightA<COUNT>(F, h1, frame, -wi, w) : sky<COUNT>(F, wi, w);
        V3 TF = T * (mat0.albedo * (cosTheta / u.pi));
        result = result + TF * Li;
        if (!hit1) break;
        T = TF * u.giScaleAnalytic;
        h = h1;
        seed = (int)((uint32_t)seed * 131u + 17u);
    }
    return result;
}

template <bool COUNT>
RT_DEV V3 shadeGlass(const Frag &F, const Hit &h, V3 wo, const MaterialProps &mat, int frame, Work &w) {   // :576-663
    const RtUniforms &u = *F.u;
    V3 N = normalize(h.n);
    V3 V = normalize(wo);
    V3 I = -V;
    float ior = mat.ior;
    float eta = 1.0f / fmaxr(ior, 1.0001f);
    const float distortionStrength = 0.45f;
    V3 camPos = ld3(u.camPos);
    V3 R = reflect(I, N);
    V3 reflectEnv = sky<COUNT>(F, R, w);
    V3 reflectLocal = reflectEnv;
    {
        Hit hRefl;
        if (traceScene<COUNT>(F, h.p + R * u.eps, R, false, true, hRefl, w, true)) {
            V3 V2v = normalize(camPos - hRefl.p);
            reflectLocal = directLightA<COUNT>(F, hRefl, frame, V2v, w);
        }
    }
    V3 reflectCol = mix(reflectEnv, reflectLocal, 0.4f);
    V3 straightCol;
    {
        Hit hS;
        if (traceScene<COUNT>(F, h.p + I * u.eps, I, false, true, hS, w, true)) {
            V3 V2v = normalize(camPos - hS.p);
            straightCol = directLightA<COUNT>(F, hS, frame, V2v, w);
        } else straightCol = sky<COUNT>(F, I, w);
    }
    float cosTheta = clampr(dot(-I, N), 0.0f, 1.0f);
    float k = 1.0f - eta * eta * (1.0f - cosTheta * cosTheta);
    V3 refrCol = straightCol;
    if (distortionStrength > 0.0f && k > 0.0f) {
        V3 T_phys = normalize(refract(I, N, eta));
        V3 T = normalize(mix(I, T_phys, distortionStrength));
        Hit hR;
        V3 bentCol;
        if (traceScene<COUNT>(F, h.p + T * u.eps, T, false, true, hR, w, true)) {
            V3 V2v = normalize(camPos - hR.p);
            bentCol = directLightA<COUNT>(F, hR, frame, V2v, w);
        } else bentCol = sky<COUNT>(F, T, w);
        refrCol = mix(straightCol, bentCol, distortionStrength);
    }
    refrCol = refrCol * mat.albedo;
    float F0 = powr((ior - 1.0f) / (ior + 1.0f), 2.0f);
    float fresnel = F0 + (1.0f - F0) * powr(1.0f - cosTheta, 5.0f);
    return mix(refrCol, reflectCol, fresnel);
}

template <bool COUNT>
RT_DEV V3 shadeMirror(const Frag &F, const Hit &h, V3 wo, const MaterialProps &mat, int frame, Work &w) {   // :675-708
    const RtUniforms &u = *F.u;
    V3 N = normalize(h.n);
    V3 I = -normalize(wo);
    V3 R = reflect(I, N);
    V3 org = h.p + R * u.eps;
    Hit h2;
    bool hit2 = traceScene<COUNT>(F, org, R, true, true, h2, w, true);
    V3 col;
    if (hit2) {
        col = directLightA<COUNT>(F, h2, frame, -R, w);
        if (u.enableGI == 1) {
            int giSeed = (int)((uint32_t)frame * 131u + 17u);
            col = col + u.giScaleAnalytic * oneBounceGIAnalytic<COUNT>(F, h2, frame, giSeed, w);
        }
    } else {
        col = (u.useEnvMap == 1) ? texture_cube<COUNT>(*F.sc, R, w) * u.envIntensity : sky<COUNT>(F, R, w);
    }
    return col * mat.albedo;
}

template <bool COUNT>
RT_DEV float computeAO_A(const Frag &F, const Hit &h, int frame, Work &w) {   // :721-757, analytic branch
    const RtUniforms &u = *F.u;
    V3 N = normalize(h.n);
    int occludedCount = 0;
    for (int i = 0; i < u.aoSamples; ++i) {
        float ox = (float)(37 * i + 3), oy = (float)(19 * i + 11);
        V2 uu = mk2(randr(F.fcx + ox, F.fcy + ox, frame), randr(F.fcy + oy, F.fcx + oy, frame));
        V3 dir = sampleHemisphereCosine(u.pi, N, uu);
        V3 org = h.p + N * u.aoBias;
        Hit tmp;
        bool hitAny = traceScene<COUNT>(F, org, dir, true, true, tmp, w);
        if (hitAny && tmp.t < u.aoRadius) occludedCount++;
    }
    float occ = (float)occludedCount / (float)u.aoSamples;
    float ao = 1.0f - occ;
    return clampr(mixr(u.aoMin, 1.0f, ao), u.aoMin, 1.0f);
}

// One analytic-mode sample of rt.frag:118-161 on a primary hit.
template <bool COUNT>
RT_DEV V3 shadeSampleAnalytic(const Frag &F, const Hit &h, V3 V, int seed, Work &w) {
    const RtUniforms &u = *F.u;
    MaterialProps mat = getMaterial(u, h.mat);
    if (mat.type == 2) return shadeGlass<COUNT>(F, h, V, mat, seed, w);
    if (mat.type == 1) return shadeMirror<COUNT>(F, h, V, mat, seed, w);
    if (h.mat == MAT_POINTLIGHT_SPHERE) {
        V3 baseCol = ld3(u.pointLightColor) * u.pointLightIntensity;
        float d = length(h.p - ld3(u.camPos));
        float falloff = 1.0f / fmaxr(d * d * 0.25f + 1.0f, 1.0f);
        return baseCol * falloff;
    }
    V3 radiance = directLightA<COUNT>(F, h, seed, V, w);
    if (u.enableGI == 1) radiance = radiance + u.giScaleAnalytic * oneBounceGIAnalytic<COUNT>(F, h, F.frameIndex, seed, w);
    if (u.enableAO == 1) radiance = radiance * computeAO_A<COUNT>(F, h, F.frameIndex, w);
    return radiance;
}

}  // namespace rtd

"""Generates tests/golden/glsl_*.npz by executing the REFERENCE's own shaders (shaders/rt/rt.frag,
rt_present.frag, read from /root/reference at run time) on SwiftShader's software OpenGL ES 3.0
through oracle/glsl_ref.py.  Runs only in the build container (needs /root/reference and the
SwiftShader libraries inside the `kaleido` wheel); the fixtures it writes are plain data: complete inputs
(uniform block bytes, BVH arrays, cube-map faces, previous-frame accumulation) and the four render
targets the reference shader produced, as float16 bit patterns.

    python tests/golden/make_glsl_golden.py       # rewrites tests/golden/glsl_*.npz and prints agreement with the oracle

Frame f of a fixture was rendered with `prev` = the shader's own COLOR0 of frame f-1 (stored as color{f-1}),
so checkers feed every implementation the same history and compare frame by frame without drift.
"""
import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
for p in (str(ROOT), str(ROOT / "tests"), str(ROOT / "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

import opengl_raytracing_amd as rt  # noqa: E402
import oracle as orc  # noqa: E402
from glsl_ref import GlslReference  # noqa: E402


def ubytes(u):
    return np.frombuffer(bytes(u), np.uint8).copy()


def h2f(a):
    return a.view(np.float16).astype(np.float32)


def tiny_env(n, seed):
    return np.random.default_rng(seed).integers(0, 256, size=(6, n, n, 3), dtype=np.uint8)


def report(name, f, got, want):
    for k, a, b in zip(("color", "motion", "gpos", "gnrm"), got, want):
        fa, fb = h2f(a), h2f(b)
        ok = np.isfinite(fa) & np.isfinite(fb)
        d = np.where(ok, fa - fb, 0.0)
        print(f"  {name} f{f} {k:6s} bit-exact {np.mean(a == b):7.4f}  rmse {np.sqrt(np.mean(d * d)):.3e}  max {np.max(np.abs(d)):.3e}")


def run(g, name, uniform_list, nodes, tris, faces, extra=None):
    """uniform_list: RtUniforms per frame.  Renders the chain with the reference GLSL, compares with the oracle, saves."""
    d = {"uniforms": np.stack([ubytes(u) for u in uniform_list])}
    prev = None
    for f, u in enumerate(uniform_list):
        got = g.render(u, nodes, tris, faces, prev)
        want, _ = orc.render(u, nodes, tris, faces, prev)
        report(name, f, got, want)
        for k, a in zip(("color", "motion", "gpos", "gnrm"), got):
            d[f"{k}{f}"] = a
        prev = got[0]
    if nodes is not None:
        d["nodes12"], d["tris12"] = nodes, tris
    if faces is not None:
        d["env"] = faces
    if extra:
        d.update(extra)
    np.savez_compressed(HERE / f"{name}.npz", **d)
    return d


def ring_env(n, seed):
    """Random cube map whose outermost texel ring is one colour on every face: GLES 3.0 filters cube maps seamlessly across
    face borders while desktop GL (the reference never enables GL_TEXTURE_CUBE_MAP_SEAMLESS) clamps per face; with equal
    border texels both give the same value, so the fixture does not depend on that API difference."""
    f = tiny_env(n, seed)
    f[:, 0, :, :] = f[:, -1, :, :] = f[:, :, 0, :] = f[:, :, -1, :] = np.array([96, 128, 160], np.uint8)
    return f


def moving_pair(p, cam0, cam1, w, h, use_bvh=False, **kw):
    vp0 = orc.mat4_mul(orc.camera_proj(cam0), orc.camera_view(cam0))
    us = [orc.frame_uniforms(p, cam0, w, h, 0, use_bvh, **kw), orc.frame_uniforms(p, cam1, w, h, 1, use_bvh, prev_vp=vp0, **kw)]
    assert us[1].cameraMoved == 1
    return us


def main():
    g = GlslReference()
    print("GL:", g.version)
    faces = ring_env(8, 11)

    # A. analytic scene, gradient sky, defaults (BASELINE config 1 in miniature), 3 frames of TAA
    p = orc.default_render_params(); p.enableEnvMap = 0
    cam = orc.default_camera(); cam.aspect = 64 / 48
    run(g, "glsl_analytic_gradient_64x48", [orc.frame_uniforms(p, cam, 64, 48, f, False, env_loaded=False) for f in range(3)], None, None, None)

    # B. analytic scene with cube map, glass + mirror spheres on, 2 spp, point light on, 2 frames
    p = orc.default_render_params(); p.sppPerFrame = 2; p.matGlassEnabled = 1; p.matMirrorEnabled = 1; p.pointLightEnabled = 1
    cam = orc.default_camera(); cam.aspect = 48 / 36
    run(g, "glsl_analytic_materials_env_48x36", [orc.frame_uniforms(p, cam, 48, 36, f, False) for f in range(2)], None, None, faces)

    # C. analytic scene, camera moves between frame 0 and 1 (reprojection, motion vectors, disocclusion), sun on
    p = orc.default_render_params(); p.sunEnabled = 1
    cam0 = orc.default_camera(); cam0.aspect = 48 / 36
    cam1 = orc.default_camera(); cam1.aspect = 48 / 36; cam1.pos[0] += 0.25; cam1.pos[2] -= 0.1; cam1.yaw += 2.0; cam1.pitch -= 1.0
    run(g, "glsl_analytic_moving_48x36", moving_pair(p, cam0, cam1, 48, 36), None, None, faces)

    # D. toggles: GI off, AO off, TAA off, jitter off, sky light off, looking up into the cube map (direct lookups on +Y and side faces)
    p = orc.default_render_params(); p.enableGI = 0; p.enableAO = 0; p.enableTAA = 0; p.enableJitter = 0; p.skyEnabled = 0
    cam = orc.default_camera(); cam.aspect = 48 / 36; cam.pitch = 35.0; cam.yaw = -60.0
    run(g, "glsl_analytic_toggles_48x36", [orc.frame_uniforms(p, cam, 48, 36, f, False) for f in range(2)], None, None, faces)

    # E. present pass over B's last frame: SVGF on, SVGF off, motion view (over C's moving frame)
    dB = dict(np.load(HERE / "glsl_analytic_materials_env_48x36.npz"))
    dC = dict(np.load(HERE / "glsl_analytic_moving_48x36.npz"))
    pres = {}
    p = orc.default_render_params()
    for tag, svgf, show, src in (("svgf", 1, False, dB), ("plain", 0, False, dB), ("motion", 1, True, dC), ("svgf_moving", 1, False, dC)):
        targets = [src[f"{k}1"] for k in ("color", "motion", "gpos", "gnrm")]
        p.enableSVGF = svgf
        pp = rt.make_present_params(p, show, 48, 36)
        got = g.present(pp, targets)
        want = orc.present(pp, targets)
        diff = np.abs(got.astype(np.int32) - want.astype(np.int32))
        print(f"  present {tag}: bit-exact {np.mean(got == want):.4f}  max |d| {diff.max()}")
        pres[f"pp_{tag}"] = ubytes(pp)
        pres[f"rgba_{tag}"] = got
        for k, a in zip(("color", "motion", "gpos", "gnrm"), targets):
            pres[f"{k}_{tag}"] = a
    np.savez_compressed(HERE / "glsl_present_48x36.npz", **pres)

    # F. BVH primitives, one call per case (the traversal loops themselves do not run on SwiftShader 4.1, see oracle/glsl_ref.py)
    v, fidx = rt.meshgen.bunny_standin(3)
    nodes, tris = orc.build_bvh(orc.gather_triangles(v, fidx))
    n = min(nodes.shape[0], tris.shape[0], 1280)
    rng = np.random.default_rng(5)
    rays = np.zeros((n, 8), np.float32)
    for i in range(n):
        # aim at a random point of triangle i (hit), of a neighbouring triangle (near miss) or at node i's box
        tri = tris[(i + (0 if i % 3 else 1)) % n]
        w = rng.dirichlet(np.ones(3)) if i % 5 else np.array([1.0, 0.0, 0.0])       # every 5th: exactly a vertex
        target = tri[0:3] + w[1] * tri[4:7] + w[2] * tri[8:11]
        if i % 7 == 0:
            target = 0.5 * (nodes[i, 0:3] + nodes[i, 4:7]) + rng.normal(0, 0.3, 3) * (nodes[i, 4:7] - nodes[i, 0:3])
        ro = np.array([-2.0, 1.5, 1.0]) + rng.normal(0, 0.5, 3)
        d = target - ro
        dist = np.linalg.norm(d)
        rays[i, 0:3] = ro
        rays[i, 3] = 1e30 if i % 4 else dist * 0.999      # tMax: every 4th just short of the hit
        rays[i, 4:7] = (d / dist).astype(np.float32)
        if i % 11 == 0:
            rays[i, 4 + i % 3] = 0.0                         # axis-parallel component: rdInv = inf
    eps = orc.frame_uniforms(orc.default_render_params(), orc.default_camera(), 8, 8, 0, False).eps
    o0, o1, o2 = g.bvh_kat(nodes, tris, rays, eps)
    u = orc.frame_uniforms(orc.default_render_params(), orc.default_camera(), 8, 8, 0, True, n, n)
    agree = {"aabb_hit": 0, "aabb_t": 0, "tri_hit": 0, "tri_t": 0, "tri_n": 0, "hits": 0}
    for i in range(n):
        a = orc.aabb_hit(rays[i, 0:3], rays[i, 4:7], nodes[i, 0:3], nodes[i, 4:7])
        t = orc.tri_hit(u, rays[i, 0:3], rays[i, 4:7], tris[i], rays[i, 3])
        agree["aabb_hit"] += a[0] == o0[i, 0]
        agree["aabb_t"] += bool(a[1] == o0[i, 1] and a[2] == o0[i, 2]) or bool(np.isnan(a[1:3]).any() and np.isnan(o0[i, 1:3]).any())
        agree["tri_hit"] += t[0] == o1[i, 0]
        if t[0] and o1[i, 0]:
            agree["hits"] += 1
            agree["tri_t"] += t[1] == o1[i, 1]
            agree["tri_n"] += bool((t[2:5] == o2[i, 0:3]).all())
    print("  bvh primitives:", n, "cases;", {k: int(v) for k, v in agree.items()})
    np.savez_compressed(HERE / "glsl_bvh_kat.npz", nodes12=nodes[:n], tris12=tris[:n], rays=rays, eps=np.float32(eps), o0=o0, o1=o1, o2=o2)

    # G. the BVH shading branch (rt.frag:92-106: directLightBVH + oneBounceGIBVH + computeAO) for synthetic hits, empty BVH
    W = 32
    nh = 512
    rng = np.random.default_rng(23)
    hits = np.zeros((nh, 12), np.float32)
    for i in range(nh):
        nrm = rng.normal(size=3); nrm /= np.linalg.norm(nrm)
        if i % 3 == 0:
            nrm = np.array([0.0, 1.0, 0.0]) + rng.normal(0, 0.2, 3); nrm /= np.linalg.norm(nrm)   # mostly facing the disk light
        vdir = nrm + rng.normal(0, 0.7, 3); vdir /= np.linalg.norm(vdir)
        hits[i, 0:3] = np.array([-2.0, 1.5, 0.0]) + rng.normal(0, 1.0, 3)
        hits[i, 3], hits[i, 7] = (i % W) + 0.5, (i // W) + 0.5
        hits[i, 4:7] = nrm * (1.0 if i % 4 else 2.5)          # un-normalised normals too: the shader normalises (rt_lighting.glsl:406)
        hits[i, 8:11] = vdir
        hits[i, 11] = float(i % 7)
    shade = {"hits": hits, "env": faces, "width": np.int32(W)}
    for tag, kw in (("default", {}), ("disk_light_only", dict(sunEnabled=0, pointLightEnabled=0, skyEnabled=0)), ("no_env_gi_only", dict(enableEnvMap=0, enableAO=0))):
        p = orc.default_render_params(); p.sppPerFrame = 2
        for k, val in kw.items():
            setattr(p, k, val)
        u = orc.frame_uniforms(p, orc.default_camera(), W, nh // W, 3, True, 0, 0, env_loaded=(p.enableEnvMap == 1))
        got = g.shade_bvh_hits(u, faces if p.enableEnvMap == 1 else None, hits, W)
        want = orc.shade_bvh_hits(u, faces if p.enableEnvMap == 1 else None, hits)
        rel = np.abs(got - want) / np.maximum(np.abs(want), 1e-3)
        print(f"  bvh shade {tag}: bit-exact {np.mean(got == want):.4f}  max rel {rel.max():.3e}  median rel {np.median(rel):.3e}  mean radiance {want.mean():.4f}")
        shade[f"u_{tag}"] = ubytes(u)
        shade[f"rad_{tag}"] = got
    np.savez_compressed(HERE / "glsl_bvh_shade_kat.npz", **shade)


if __name__ == "__main__":
    main()

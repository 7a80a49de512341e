"""Generates tests/golden/glsl_*.npz by executing the REFERENCE's own shaders (shaders/rt/rt.frag,
rt_present.frag, read from /root/reference at run time) on SwiftShader's software OpenGL ES 3.0
through oracle/glsl_ref.py.  Runs only in the build container (needs /root/reference and the
SwiftShader libraries inside the `kaleido` wheel); the fixtures it writes are plain data: complete inputs
(uniform block bytes, BVH arrays, cube-map faces, previous-frame accumulation) and the four render
targets the reference shader produced, as float16 bit patterns.

    python tests/golden/make_glsl_golden.py           # rewrites tests/golden/glsl_*.npz and prints agreement with the oracle
    python tests/golden/make_glsl_golden.py H I J     # only the named sections (A-G: round 1, H-I: the BVH traversal loops, J: TAA weight regimes)

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


def crate_mesh(nx=8, ny=8, seed=7, layers=(0.0, -4.0)):
    """Height field over an integer grid, heights in {0,1,2}, two layers, cells split along alternating diagonals: every
    coordinate is a small integer, so Moller-Trumbore along axis directions is exact and rays through shared edges hit two
    triangles at bit-equal t -- the tie rule of triHit / traceBVH (`tt > tMax` rejects, so the LATER triangle wins,
    rt_bvh.glsl:166,218) and the equal-tmin push order (`tminL < tminR`, :233) decide the returned normal."""
    rng = np.random.default_rng(seed)
    out = []
    for layer, z0 in enumerate(layers):
        h = rng.integers(0, 3, size=(nx + 1, ny + 1)).astype(np.float32) + np.float32(z0)
        for x in range(nx):
            for y in range(ny):
                def P(i, j):
                    return np.array([x + i, y + j, h[x + i, y + j]], np.float32)
                if (x + y + layer) % 2 == 0:
                    cell = [(P(0, 0), P(1, 0), P(1, 1)), (P(0, 0), P(1, 1), P(0, 1))]
                else:
                    cell = [(P(0, 0), P(1, 0), P(0, 1)), (P(1, 0), P(1, 1), P(0, 1))]
                for a, b, c in cell:
                    out.append(np.concatenate([a, b - a, c - a]))
    return np.array(out, np.float32)


def trace_kat(g):
    """H. traceBVH / traceBVHShadow executed ray by ray (oracle/glsl_ref.py bvh_trace_kat)."""
    d = {}
    eps_inf = orc.frame_uniforms(orc.default_render_params(), orc.default_camera(), 8, 8, 0, False)
    # H1: exact-arithmetic mesh, axis rays through edges / cell interiors (ties, rdInv = +-inf), up and down
    nodes, tris = orc.build_bvh(crate_mesh())
    rays = []
    for x in range(8):
        for y in range(8):
            for fx, fy in ((0.5, 0.5), (0.25, 0.25), (0.75, 0.75), (0.25, 0.75), (0.75, 0.25), (0.25, 0.5), (0.0, 0.0), (0.5, 0.0), (0.0, 0.25)):
                rays.append(((x + fx, y + fy, 8.0), (0.0, 0.0, -1.0)))
                rays.append(((x + fx, y + fy, -12.0), (0.0, 0.0, 1.0)))
    for k in range(64):                         # sideways: two infinite slab axes at once
        rays.append(((-3.0, 0.125 + 0.25 * (k % 31), -3.875 + 0.25 * (k // 2)), (1.0, 0.0, 0.0)))
        rays.append(((0.375 + 0.25 * (k % 29), 11.0, -3.625 + 0.25 * (k // 2)), (0.0, -1.0, 0.0)))
    r8 = np.zeros((len(rays), 8), np.float32)
    for i, (o, dd) in enumerate(rays):
        r8[i, 0:3], r8[i, 4:7] = o, dd
        r8[i, 3] = (9.0, 6.5, 14.0, 2.0)[i % 4]
    d["crate"] = (nodes, tris, r8)
    # H2: a generic mesh (bunny stand-in, 320 triangles, depth-7 tree), rays in general position + axis-parallel components
    v, fidx = rt.meshgen.bunny_standin(2)
    nodes, tris = orc.build_bvh(orc.gather_triangles(v, fidx))
    rng = np.random.default_rng(29)
    n = 1536
    centre = 0.5 * (nodes[0, 0:3] + nodes[0, 4:7])
    ro = (centre + rng.normal(0, 1, (n, 3)) * 1.2).astype(np.float32)
    ro[: n // 8] = (centre + rng.normal(0, 0.1, (n // 8, 3))).astype(np.float32)          # origins inside the mesh
    tgt = tris[rng.integers(tris.shape[0], size=n), 0:3] + rng.normal(0, 0.15, (n, 3))
    rd = tgt - ro
    rd = (rd / np.linalg.norm(rd, axis=1, keepdims=True)).astype(np.float32)
    for i in range(0, n, 13):
        rd[i, i % 3] = 0.0                                                                # rdInv = inf on one axis
    r8 = np.zeros((n, 8), np.float32)
    r8[:, 0:3], r8[:, 4:7] = ro, rd
    r8[:, 3] = rng.uniform(0.2, 4.0, n).astype(np.float32)
    d["bunny"] = (nodes, tris, r8)
    save = {"eps": np.float32(eps_inf.eps), "inf": np.float32(eps_inf.inf)}
    for tag, (nodes, tris, r8) in d.items():
        o0, o1, o2 = g.bvh_trace_kat(nodes, tris, r8, eps_inf.eps, eps_inf.inf)
        u = orc.frame_uniforms(orc.default_render_params(), orc.default_camera(), 8, 8, 0, True, nodes.shape[0], tris.shape[0])
        nan_slab = np.zeros(r8.shape[0], np.uint8)
        ties = np.zeros(r8.shape[0], np.uint8)
        agree = {"hit": 0, "t_exact": 0, "t_close": 0, "normal": 0, "occ": 0, "hits": 0}
        for i in range(r8.shape[0]):
            o, dd = r8[i, 0:3], r8[i, 4:7]
            # 0 * inf = NaN in a slab: the ray lies IN a box plane of an axis it does not move along.  GLSL leaves min/max of a
            # NaN to the driver (SwiftShader: SSE minps/maxps; the oracle models v_min_f32 / v_max_f32) -- recorded, not compared.
            for ax in range(3):
                if dd[ax] == 0.0 and (np.any(nodes[:, ax] == o[ax]) or np.any(nodes[:, 4 + ax] == o[ax])):
                    nan_slab[i] = 1
            hit, t, pp, nn, _ = orc.trace_bvh(u, nodes, tris, o, dd)
            occ = orc.trace_bvh_shadow(u, nodes, tris, o, dd, float(r8[i, 3]))
            if hit:
                same = sum(1 for k in range(tris.shape[0]) if (lambda r: r[0] and r[1] == np.float32(t))(orc.tri_hit(u, o, dd, tris[k], 1e30)))
                ties[i] = min(same, 255)
            if nan_slab[i]:
                continue
            agree["hit"] += bool(o0[i, 0]) == hit
            agree["occ"] += bool(o0[i, 2]) == occ
            if hit and o0[i, 0]:
                agree["hits"] += 1
                agree["t_exact"] += o0[i, 1] == np.float32(t)
                agree["t_close"] += abs(o0[i, 1] - t) <= 2e-6 * abs(t)
                agree["normal"] += np.abs(o2[i, 0:3] - nn).max() <= 1e-5
        print(f"  trace kat {tag}: {r8.shape[0]} rays, {int(nan_slab.sum())} with a NaN slab (not compared), {int((ties > 1).sum())} with >= 2 triangles at the "
              f"winning t; of the rest: {agree}")
        save.update({f"{tag}_nodes12": nodes, f"{tag}_tris12": tris, f"{tag}_rays": r8, f"{tag}_o0": o0, f"{tag}_o1": o1, f"{tag}_o2": o2,
                     f"{tag}_nan_slab": nan_slab, f"{tag}_ties": ties})
    np.savez_compressed(HERE / "glsl_bvh_trace_kat.npz", **save)


def bvh_frames(g, faces):
    """I. whole BVH frames through rt.frag (uUseBVH = 1): primary traversal, shadow / bounce / AO rays, TAA."""
    v, fidx = rt.meshgen.bunny_standin(3)                       # 1 280 triangles, 511 nodes
    nodes, tris = orc.build_bvh(orc.gather_triangles(v, fidx))
    # close-up camera (mesh ~45 % of the frame), 2 spp, defaults (GI + AO + sun + sky + point + env), 3 frames of TAA
    W, H = 64, 48
    p = orc.default_render_params(); p.sppPerFrame = 2
    cam = orc.default_camera(); cam.pos[0], cam.pos[1], cam.pos[2], cam.yaw, cam.pitch, cam.aspect = -2.0, 1.5, 1.0, -90.0, 0.0, W / H
    run(g, "glsl_bvh_closeup_64x48", [orc.frame_uniforms(p, cam, W, H, f, True, nodes.shape[0], tris.shape[0]) for f in range(3)], nodes, tris, faces)
    # camera moves between frames 0 and 1: motion vectors on hits, reprojection, the (4,4) marker on misses; 1 spp, GI off
    p = orc.default_render_params(); p.enableGI = 0
    cam1 = orc.default_camera(); cam1.pos[0], cam1.pos[1], cam1.pos[2], cam1.yaw, cam1.pitch, cam1.aspect = -1.9, 1.55, 1.05, -91.5, -1.0, W / H
    run(g, "glsl_bvh_moving_64x48", moving_pair(p, cam, cam1, W, H, True, node_count=nodes.shape[0], tri_count=tris.shape[0]), nodes, tris, faces)
    # 5 120 triangles (depth-12 tree), reference default camera looking at the mesh from further away, 1 spp, gradient sky
    v, fidx = rt.meshgen.bunny_standin(4)
    nodes, tris = orc.build_bvh(orc.gather_triangles(v, fidx))
    p = orc.default_render_params(); p.enableEnvMap = 0
    cam = orc.default_camera(); cam.pos[0], cam.pos[1], cam.pos[2], cam.yaw, cam.pitch, cam.aspect = -2.0, 1.6, 2.2, -90.0, -3.0, W / H
    run(g, "glsl_bvh_5k_gradient_64x48", [orc.frame_uniforms(p, cam, W, H, f, True, nodes.shape[0], tris.shape[0], env_loaded=False) for f in range(2)],
        nodes, tris, None)


def taa_regimes(g, faces):
    """J. 34 consecutive frames of one static view (BVH scene, 1 spp): resolveTAA's still branch switches its history weight from
    uTaaHistoryMinWeight (0.85) to Avg (0.92) at uFrameIndex 8 and to Max (0.96) at 32 (rt_taa.glsl:91-104).  The whole chain runs
    on the reference GLSL; the fixture keeps the frames around the switches with the history each of them read."""
    v, fidx = rt.meshgen.bunny_standin(2)
    nodes, tris = orc.build_bvh(orc.gather_triangles(v, fidx))
    W, H = 48, 36
    p = orc.default_render_params()
    cam = orc.default_camera(); cam.pos[0], cam.pos[1], cam.pos[2], cam.yaw, cam.pitch, cam.aspect = -2.0, 1.5, 1.0, -90.0, 0.0, W / H
    keep = (0, 1, 7, 8, 9, 31, 32, 33)
    d = {"frames": np.array(keep, np.int32), "nodes12": nodes, "tris12": tris, "env": faces}
    prev = None
    for f in range(34):
        u = orc.frame_uniforms(p, cam, W, H, f, True, nodes.shape[0], tris.shape[0])
        got = g.render(u, nodes, tris, faces, prev)
        if f in keep:
            want, _ = orc.render(u, nodes, tris, faces, prev)
            report("glsl_bvh_taa_regimes_48x36", f, got, want)
            d[f"uniforms{f}"] = ubytes(u)
            if prev is not None:
                d[f"prev{f}"] = prev
            for k, a in zip(("color", "motion", "gpos", "gnrm"), got):
                d[f"{k}{f}"] = a
        prev = got[0]
    np.savez_compressed(HERE / "glsl_bvh_taa_regimes_48x36.npz", **d)


def main():
    g = GlslReference()
    print("GL:", g.version)
    faces = ring_env(8, 11)
    only = set(a.upper() for a in sys.argv[1:])
    if only:
        if "H" in only:
            trace_kat(g)
        if "I" in only:
            bvh_frames(g, faces)
        if "J" in only:
            taa_regimes(g, faces)
        if only <= {"H", "I", "J"}:
            return

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

    # F. BVH primitives, one call per case (the traversal loops: sections H and I)
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

    trace_kat(g)
    bvh_frames(g, faces)
    taa_regimes(g, faces)


if __name__ == "__main__":
    main()

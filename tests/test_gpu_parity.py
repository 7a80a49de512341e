"""Parity tests proper (run on a real MI355X with `-m gpu`): the HIP path, called through the C ABI
of librt_mi355.so, against the CPU oracle on the same inputs.

Bars (north_star): integer / index work bit-exact (RNG, BVH arrays, fp16 bit patterns of the
deterministic float model); floating point: COLOR0 RMSE < 1e-4 after fp16 rounding.  The float
model is designed to be bit-reproducible, so the tests first demand bit-equality and report the
RMSE / outlier figures if that ever fails.
"""
import numpy as np
import pytest

import opengl_raytracing_amd as rt
import scenes

pytestmark = pytest.mark.gpu

RMSE_TOL = 1e-4  # BASELINE.json north_star: "pixel RMSE < 1e-4"


@pytest.fixture(scope="module")
def ren():
    r = rt.Renderer(count_work=True, pipeline=rt.RT_PIPELINE_MEGAKERNEL)
    yield r
    r.close()


def _assert_targets_equal(got, want, orc, what):
    names = ["color", "motion", "gpos", "gnrm"]
    for g, w, n in zip(got, want, names):
        st = orc.compare(g, w)
        assert st["rmse"] < RMSE_TOL, f"{what}/{n}: {st}"
        assert st["bit_diff"] == 0, f"{what}/{n}: not bit-identical: {st}"


def test_device_float_model_bit_exact(ren, orc):
    rng = np.random.default_rng(1)
    L = orc.lib()
    x = np.concatenate([rng.uniform(-7, 7, 4000), [0.0, -0.0, np.pi / 4, np.pi / 2, np.pi, 2 * np.pi, 1e-8]]).astype(np.float32)
    for op, fn in ((0, L.orc_sin), (1, L.orc_cos)):
        want = np.array([fn(float(v)) for v in x], np.float32).view(np.uint32)
        assert np.array_equal(ren.debug_eval(op, x), want)
    t = rng.uniform(-160, 140, 4000).astype(np.float32)
    want = np.array([L.orc_exp2(float(v)) for v in t], np.float32).view(np.uint32)
    assert np.array_equal(ren.debug_eval(2, t), want)
    p = np.exp(rng.uniform(-90, 80, 4000)).astype(np.float32)
    want = np.array([L.orc_log2(float(v)) for v in p], np.float32).view(np.uint32)
    assert np.array_equal(ren.debug_eval(3, p), want)
    base = np.concatenate([rng.uniform(0, 1, 4000), [0.0, 1.0]]).astype(np.float32)
    expo = rng.choice(np.array([2.0, 5.0, 16.0, 32.0, 48.0, 256.0], np.float32), base.size)
    want = np.array([L.orc_pow(float(a), float(b)) for a, b in zip(base, expo)], np.float32).view(np.uint32)
    assert np.array_equal(ren.debug_eval(4, base, expo), want)
    # division / sqrt must be correctly rounded on the device (numpy float32 is)
    a = rng.uniform(-100, 100, 4000).astype(np.float32)
    b = rng.uniform(0.01, 100, 4000).astype(np.float32)
    assert np.array_equal(ren.debug_eval(7, a, b), (a / b).view(np.uint32))
    assert np.array_equal(ren.debug_eval(8, b), np.sqrt(b).view(np.uint32))
    assert np.array_equal(ren.debug_eval(9, b), (np.float32(1.0) / np.sqrt(b)).view(np.uint32))


def test_device_fp16_rounding_and_rng(ren, orc):
    rng = np.random.default_rng(2)
    L = orc.lib()
    x = np.concatenate([rng.uniform(-70000, 70000, 3000), rng.uniform(-1e-4, 1e-4, 3000), rng.uniform(0, 2, 3000),
                        [0.0, 65504.0, 65519.9, 65520.0, 1e-8, 6.1e-5, 5.96e-8, 2.98e-8]]).astype(np.float32)
    want = np.array([L.orc_f32_to_f16(float(v)) for v in x], np.uint32)
    assert np.array_equal(ren.debug_eval(5, x), want)
    assert np.array_equal(want.astype(np.uint16), x.astype(np.float16).view(np.uint16))  # oracle == IEEE RNE
    px = rng.uniform(0, 4000, 3000).astype(np.float32)
    py = rng.uniform(0, 4000, 3000).astype(np.float32)
    fr = rng.integers(0, 100000, 3000).astype(np.float32)
    want = np.array([L.orc_rand_bits(float(a), float(b), int(c)) for a, b, c in zip(px, py, fr)], np.uint32)
    assert np.array_equal(ren.debug_eval(6, px, py, fr), want)


def test_traversal_matches_oracle_ray_by_ray(ren, orc):
    nodes, tris = scenes.bunny_bvh(3)   # 1280 triangles
    ren.upload_bvh(nodes, tris)
    u = rt.frame_uniforms(rt.default_render_params(), rt.default_camera(), 64, 64, 0, True, nodes.shape[0], tris.shape[0])
    rng = np.random.default_rng(5)
    n = 3000
    target = np.array([-2.0, 1.5, 0.0], np.float32) + rng.uniform(-0.6, 0.6, (n, 3)).astype(np.float32)
    org = (np.array([-2.0, 1.5, 0.0], np.float32) + rng.normal(size=(n, 3)).astype(np.float32) * 2.0).astype(np.float32)
    d = (target - org).astype(np.float32)
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    d[:8] = np.array([[0, 0, -1], [0, 0, 1], [1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, -1], [1, 0, 0]], np.float32)  # 1/0 = inf slabs
    got = ren.debug_trace(0, org, d)
    hits = 0
    for i in range(n):
        hit, t, p, nn, _ = orc.trace_bvh(u, nodes, tris, org[i], d[i])
        if hit:
            hits += 1
            assert got[i, 0].view(np.uint32) == np.float32(t).view(np.uint32), i
            assert np.array_equal(got[i, 1:4].view(np.uint32), p.view(np.uint32)), i
            assert np.array_equal(got[i, 4:7].view(np.uint32), nn.view(np.uint32)), i
        else:
            assert got[i, 0] == np.float32(1e30), i
    assert hits > n // 4
    tmax = rng.uniform(0.2, 4.0, n).astype(np.float32)
    got = ren.debug_trace(1, org, d, tmax)
    want = np.array([orc.trace_bvh_shadow(u, nodes, tris, org[i], d[i], tmax[i]) for i in range(n)], np.float32)
    assert np.array_equal(got[:, 0], want)


@pytest.mark.parametrize("qn", [0, 2, "million", "fused", "implicit", "implicit_q", "implicit_million"])
def test_wavefront_traversal_kernels_ray_by_ray_on_adversarial_rays(orc, monkeypatch, qn):
    """rt_debug_trace kinds 2 / 3: arbitrary rays through the PRODUCTION traversal kernels (k_trace: persistent launch, refill scheduler, 4-wide any-hit
    nodes -- exact and, with RT_QNODES=2, quantised), answer by answer against the oracle's restatement of traceBVH / traceBVHShadow.  Besides random rays:
    axis-parallel rays (1/0 = inf slabs) whose origin coordinates sit EXACTLY on planes of node boxes, rays that start on triangle vertices and edge
    midpoints (what shadow and AO rays do), rays aimed at box corners, rays with denormal-size direction components.  The quantised form must return the
    same bits: its inner boxes only ever ADD candidates, the exact box test at the leaf decides."""
    if qn == "fused":                   # round 5: the closest-hit launches on the fused records (RT_FUSED=1: the reference's order, two binary steps per round trip)
        monkeypatch.setenv("RT_FUSED", "1")
        qn = 0
    if qn in ("implicit", "implicit_q", "implicit_million"):
        # round 5, RT_IMPLICIT=1: records without child references (the mesh's leaves all sit at depth 10; the 1 M-triangle scene's at depth 17) -- 48 bytes for the
        # closest-hit launches, 96 bytes for the any-hit launches, 48 bytes for their quantised form (implicit_q: forced on; implicit_million: chosen by size)
        monkeypatch.setenv("RT_IMPLICIT", "1")
        qn = {"implicit": 0, "implicit_q": 2, "implicit_million": "million"}[qn]
    if qn == "million":                 # the 1 M-triangle scene with the form rt_upload_bvh chooses for it by itself (quantised: 9.8 MB of exact nodes)
        monkeypatch.delenv("RT_QNODES", raising=False)
        v_, f_ = rt.meshgen.million_triangle_scene()
        nodes, tris = rt.build_bvh(rt.gather_triangles(v_, f_, np.eye(4, dtype=np.float32).reshape(-1)))
    else:
        monkeypatch.setenv("RT_QNODES", str(qn))
        nodes, tris = scenes.bunny_bvh(4)   # 5120 triangles, depth-12 tree
    u = rt.frame_uniforms(rt.default_render_params(), rt.default_camera(), 64, 64, 0, True, nodes.shape[0], tris.shape[0])
    rng = np.random.default_rng(11)
    lo, hi = nodes[:, 0:3], nodes[:, 4:7]
    centre = ((lo[0] + hi[0]) * 0.5).astype(np.float32)
    ext = float((hi[0] - lo[0]).max())
    f32 = np.float32

    def unit(v):
        v = v.astype(np.float32)
        return (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32)

    O, D = [], []
    n = 6000
    # (1) random rays towards the mesh
    o = (centre + rng.normal(size=(n, 3)) * ext).astype(f32)
    O.append(o); D.append(unit(centre + rng.uniform(-0.4, 0.4, (n, 3)) * ext - o))
    # (2) axis-parallel rays lying exactly in planes of node boxes (two coordinates of the origin taken from box corners of random nodes)
    for axis in range(3):
        k = rng.integers(0, nodes.shape[0], n)
        corner = np.where(rng.random((n, 3)) < 0.5, lo[k], hi[k]).astype(f32)
        o = corner.copy()
        o[:, axis] = (centre[axis] + np.where(rng.random(n) < 0.5, -1.0, 1.0) * ext * 1.5).astype(f32)
        d = np.zeros((n, 3), f32)
        d[:, axis] = -np.sign(o[:, axis] - centre[axis])
        O.append(o); D.append(d)
    # (3) rays leaving triangle vertices and edge midpoints
    k = rng.integers(0, tris.shape[0], n)
    v0, e1, e2 = tris[k, 0:3], tris[k, 4:7], tris[k, 8:11]
    start = np.where((rng.random(n) < 0.5)[:, None], v0, (v0 + f32(0.5) * e1).astype(f32)).astype(f32)
    O.append(start); D.append(unit(rng.normal(size=(n, 3))))
    # (4) rays aimed exactly at corners of node boxes
    k = rng.integers(0, nodes.shape[0], n)
    corner = np.where(rng.random((n, 3)) < 0.5, lo[k], hi[k]).astype(f32)
    o = (centre + unit(rng.normal(size=(n, 3))) * ext * 2.0).astype(f32)
    O.append(o); D.append(unit(corner - o))
    # (5) direction components of denormal size and exact zeros mixed in
    o = (centre + rng.normal(size=(n, 3)) * ext).astype(f32)
    d = unit(centre - o)
    tiny = rng.integers(0, 3, n)
    d[np.arange(n), tiny] = np.where(rng.random(n) < 0.5, f32(1e-41), f32(-0.0))
    O.append(o); D.append(d)
    org, dirs = np.concatenate(O).astype(f32), np.concatenate(D).astype(f32)
    N = org.shape[0]
    tmax = rng.uniform(0.05, 3.0, N).astype(f32) * f32(ext)
    with rt.Renderer(pipeline=rt.RT_PIPELINE_WAVEFRONT) as r:
        r.upload_bvh(nodes, tris)
        closest = r.debug_trace(2, org, dirs)
        anyhit = r.debug_trace(3, org, dirs, tmax)
        ref_any = r.debug_trace(1, org, dirs, tmax)          # the megakernel's walk of the same question
    hits = occluded = 0
    for i in range(N):
        hit, t, p, nn, _ = orc.trace_bvh(u, nodes, tris, org[i], dirs[i])
        if hit:
            hits += 1
            assert closest[i, 0].view(np.uint32) == np.float32(t).view(np.uint32), (qn, i)
            tri = int(closest[i, 1])
            g = np.cross(tris[tri, 4:7].astype(np.float64), tris[tri, 8:11].astype(np.float64))
            g /= np.linalg.norm(g)
            assert abs(abs(float(np.dot(g, nn.astype(np.float64)))) - 1.0) < 1e-4, (qn, i, tri)     # the triangle the oracle's normal belongs to
        else:
            assert closest[i, 0] == np.float32(1e30), (qn, i)
        occ = orc.trace_bvh_shadow(u, nodes, tris, org[i], dirs[i], tmax[i])
        occluded += occ
        assert bool(anyhit[i, 0]) == occ, (qn, i)
    assert np.array_equal(anyhit[:, 0], ref_any[:, 0])
    assert hits > N // 8 and occluded > N // 16


@pytest.mark.parametrize("env", [None, "Sky_16", "tiny"])
def test_analytic_scene_frames(ren, orc, env):
    """BASELINE config 1 (analytic 256x256) through the HIP path, frames 0..3 with TAA history."""
    W = H = 256
    faces = None if env is None else (scenes.tiny_env() if env == "tiny" else scenes.env_faces(env))
    ren.upload_env(faces)
    ren.resize(W, H)
    p = rt.default_render_params()
    p.enableEnvMap = 0 if env is None else 1
    cam = scenes.camera("default", aspect=1.0)
    prev = None
    for frame in range(4):
        u = rt.frame_uniforms(p, cam, W, H, frame, False)
        assert ren.frame_index == frame
        ren.reset_counters()
        ren.render_frame(u)
        got = ren.read_all()
        want, cnt = orc.render(u, env_faces=faces, prev=prev)
        _assert_targets_equal(got, want, orc, f"analytic env={env} frame={frame}")
        c = ren.counters()
        assert (c.raysAnalytic, c.envLookup, c.hitPixels) == (cnt.raysAnalytic, cnt.envLookup, cnt.hitPixels)
        prev = want[0]


@pytest.mark.parametrize("use_bvh", [False, True])
def test_cubemap_quantised_filter_mode(orc, use_bvh):
    """RtExtension.envFilter = 1 (SURVEY.md 8c's second cube-map filter model: texel coordinates rounded to 1/256 texel): HIP == oracle bit
    for bit in that mode too, on the analytic scene (megakernel) and on a BVH scene (wavefront pipeline), and the mode is not a no-op --
    the frame differs from the default model's, by less than an 8-bit step of the environment (x envIntensity) per sky lookup."""
    W, H = 128, 96
    faces = scenes.env_faces("Sky_16")
    nodes, tris = scenes.bunny_bvh(3)
    p = rt.default_render_params()
    p.sppPerFrame = 2
    cam = scenes.camera("closeup" if use_bvh else "default", aspect=W / H)
    frames = {}
    for mode in (0, 1):
        with rt.Renderer() as r:
            r.upload_bvh(nodes, tris)
            r.upload_env(faces)
            r.resize(W, H)
            r.set_extension(env_filter=mode)
            prev = None
            for f in range(2):
                u = rt.frame_uniforms(p, cam, W, H, f, use_bvh, nodes.shape[0], tris.shape[0])
                r.render_frame(u)
                got = r.read_all()
                want, _ = orc.render(u, nodes, tris, faces, prev, env_filter=mode)
                _assert_targets_equal(got, want, orc, f"env_filter={mode} bvh={use_bvh} frame={f}")
                prev = want[0]
            frames[mode] = got[0]
    d = orc.compare(frames[0], frames[1])
    assert d["bit_diff"] > 0 and d["max_abs"] < 0.05, d
    with rt.Renderer() as r, pytest.raises(rt.RtError):
        r.set_extension(env_filter=2)


@pytest.mark.parametrize("mesh", ["one_leaf", "tiny", "deep"])
def test_anyhit_sah_tree_option(orc, monkeypatch, mesh):
    """RT_ANYHIT_TREE=sah (the option lives in rt_upload_bvh, so it gets a test): the any-hit launches of the wavefront pipeline walk a
    binned-SAH 4-wide tree over the reference's leaves instead of the collapse of the median-split tree.  Any-hit answers depend only on which
    reference leaves pass their own box test, so the frames must stay bit-identical to the oracle -- on a single-leaf mesh (the option must leave
    it alone), a 20-triangle mesh (one level) and a 5 120-triangle mesh (depth-12 tree, deep SAH tree)."""
    monkeypatch.setenv("RT_ANYHIT_TREE", "sah")        # read by rt_upload_bvh
    W, H = 96, 64
    if mesh == "one_leaf":
        tris9 = np.array([[-1, 0, -1, 1, 0, -1, 0, 1.5, -1.2], [-1, 0, 1, 1, 0, 1, 0, 1.5, 0.5]], np.float32) + np.float32(0.25)
        nodes, tris = rt.build_bvh(tris9)
        assert nodes.shape[0] == 1
    else:
        nodes, tris = scenes.bunny_bvh(0 if mesh == "tiny" else 4)
    faces = scenes.tiny_env(8)
    p = rt.default_render_params()
    p.sppPerFrame = 2
    cam = scenes.camera("closeup", aspect=W / H)
    with rt.Renderer(pipeline=rt.RT_PIPELINE_WAVEFRONT) as r:
        r.upload_bvh(nodes, tris)
        r.upload_env(faces)
        r.resize(W, H)
        prev = None
        for f in range(2):
            u = rt.frame_uniforms(p, cam, W, H, f, True, nodes.shape[0], tris.shape[0])
            r.render_frame(u)
            want, _ = orc.render(u, nodes, tris, faces, prev)
            _assert_targets_equal(r.read_all(), want, orc, f"RT_ANYHIT_TREE=sah {mesh} frame={f}")
            prev = want[0]
        tr = r.traced_rays()
        assert mesh == "one_leaf" or tr.shadow > 0


@pytest.mark.parametrize("qn,mesh,sah", [(2, "one_leaf", 0), (2, "tiny", 0), (2, "deep", 0), (1, "deep", 0), (2, "deep", 1), (2, "deep", "near"), (2, "deep", "leafb4")])
def test_quantised_anyhit_nodes(orc, monkeypatch, qn, mesh, sah):
    """RT_QNODES (round 4; the default for any-hit trees beyond 4 MB of nodes, forced here on small meshes): the any-hit launches walk 64-byte nodes
    whose child boxes are bytes on the node's own grid -- supersets of the exact boxes, checked at upload in the kernel's own decode expression --
    and test a leaf's exact box in the leaf phase.  Answers depend only on which reference leaves pass their own box test, so frames stay
    bit-identical to the oracle: single-leaf mesh (no quantised tree is built), one-level tree, depth-12 tree, the seven-wave build of the
    kernel, the quantised form of the SAH any-hit tree, and the form built while another kernel option is selected (round 4's stress matrix caught the
    near-first build being handed the quantised array)."""
    monkeypatch.setenv("RT_QNODES", str(qn))           # read by rt_upload_bvh and by the renderer's launch sets
    if sah == "near":
        monkeypatch.setenv("RT_NEAR_FIRST", "1")       # a kernel build that walks the exact nodes: must be handed those, not the quantised array
    elif sah == "leafb4":
        monkeypatch.setenv("RT_LEAFB", "4")            # the quantised build wins over the leaf-group option
    elif sah:
        monkeypatch.setenv("RT_ANYHIT_TREE", "sah")
    W, H = 96, 64
    if mesh == "one_leaf":
        tris9 = np.array([[-1, 0, -1, 1, 0, -1, 0, 1.5, -1.2], [-1, 0, 1, 1, 0, 1, 0, 1.5, 0.5]], np.float32) + np.float32(0.25)
        nodes, tris = rt.build_bvh(tris9)
    else:
        nodes, tris = scenes.bunny_bvh(0 if mesh == "tiny" else 4)
    faces = scenes.tiny_env(8)
    p = rt.default_render_params()
    p.sppPerFrame = 2
    cam = scenes.camera("closeup", aspect=W / H)
    with rt.Renderer(pipeline=rt.RT_PIPELINE_WAVEFRONT) as r:
        r.upload_bvh(nodes, tris)
        r.upload_env(faces)
        r.resize(W, H)
        prev = None
        for f in range(2):
            u = rt.frame_uniforms(p, cam, W, H, f, True, nodes.shape[0], tris.shape[0])
            r.render_frame(u)
            want, _ = orc.render(u, nodes, tris, faces, prev)
            _assert_targets_equal(r.read_all(), want, orc, f"RT_QNODES={qn} sah={sah} {mesh} frame={f}")
            prev = want[0]


@pytest.mark.parametrize("mesh", ["one_leaf", "tiny", "deep", "ragged", "odd_counts", "deep_q", "odd_depth", "odd_depth_q"])
def test_implicit_closest_hit_records(orc, monkeypatch, mesh):
    """RT_IMPLICIT=1 (round 5, a measured option): when every leaf of the uploaded tree sits at one depth D a node is named by (depth, path), its children and
    the leaves' triangle records follow by arithmetic, and the closest-hit launches read 48-byte records -- the two child boxes, three loads -- instead of 64-byte
    ones; the any-hit launches read 96-byte four-child records (six loads instead of seven) or, quantised, 48-byte ones (three instead of four).  Same boxes, order and triangle tests, hence the same frames: a single leaf (nothing to do), a one-level tree, a depth-10 tree, a mesh whose leaves
    sit at two depths (rt_upload_bvh must say so and the option fall back to the explicit records), and a perfect tree whose leaves hold 5 or 6 triangles (the
    count travels in the leaf's first record)."""
    monkeypatch.setenv("RT_IMPLICIT", "1")
    if mesh.endswith("_q"):                # the any-hit launches on the quantised form of the implicit records (three loads per node visit)
        monkeypatch.setenv("RT_QNODES", "2")
        mesh = mesh[:-2]
    W, H = 96, 64
    rng = np.random.default_rng(3)
    if mesh == "one_leaf":
        tris9 = np.array([[-1, 0, -1, 1, 0, -1, 0, 1.5, -1.2], [-1, 0, 1, 1, 0, 1, 0, 1.5, 0.5]], np.float32) + np.float32(0.25)
        nodes, tris = rt.build_bvh(tris9)
    elif mesh in ("ragged", "odd_counts", "odd_depth"):
        # random small triangles around the close-up camera's target: 1100 of them split into leaves at depths 7 and 8 (1100 / 2^7 = 8.6), 1400 into a perfect
        # depth-8 tree with five or six triangles per leaf (1400 / 2^8 = 5.5), 700 into a perfect depth-7 tree (an ODD depth: the last four-wide nodes of the
        # any-hit walk hold two leaves and two absent children)
        n = {"ragged": 1100, "odd_counts": 1400, "odd_depth": 700}[mesh]
        c = (np.array([-2.0, 1.5, 0.0]) + rng.uniform(-0.5, 0.5, (n, 3))).astype(np.float32)
        tris9 = np.concatenate([c, c + rng.uniform(-0.08, 0.08, (n, 3)).astype(np.float32), c + rng.uniform(-0.08, 0.08, (n, 3)).astype(np.float32)], axis=1).astype(np.float32)
        nodes, tris = rt.build_bvh(tris9)
    else:
        nodes, tris = scenes.bunny_bvh(0 if mesh == "tiny" else 4)
    faces = scenes.tiny_env(8)
    p = rt.default_render_params()
    p.sppPerFrame = 2
    cam = scenes.camera("closeup", aspect=W / H)
    with rt.Renderer(pipeline=rt.RT_PIPELINE_WAVEFRONT) as r:
        r.upload_bvh(nodes, tris)
        info = r.scene_info()
        implicit = bool(info.flags & rt.RT_SCENE_IMPLICIT)
        assert implicit == (mesh in ("tiny", "deep", "odd_counts", "odd_depth")), (mesh, info.flags, info.implicitDepth)
        if mesh == "odd_depth":
            assert info.implicitDepth == 7
        if mesh == "deep":
            assert info.implicitDepth == 10
        r.upload_env(faces)
        r.resize(W, H)
        prev = None
        for f in range(2):
            u = rt.frame_uniforms(p, cam, W, H, f, True, nodes.shape[0], tris.shape[0])
            r.render_frame(u)
            want, _ = orc.render(u, nodes, tris, faces, prev)
            _assert_targets_equal(r.read_all(), want, orc, f"RT_IMPLICIT=1 {mesh} frame={f}")
            prev = want[0]


@pytest.mark.parametrize("mesh,inflate", [("one_leaf", False), ("tiny", False), ("deep", False), ("deep", True)])
def test_fused_closest_hit_records(orc, monkeypatch, mesh, inflate):
    """RT_FUSED=1 (round 5, a measured option): the closest-hit launches walk 128-byte records that hold the 64-byte records of a node's two children, so that the
    reference's step at a node and its step at the near child (rt_bvh.glsl:205-241) come out of one round trip; the children's own boxes are derived as the
    unions of the grandchildren's (tests/test_fused_nodes.py).  Same visiting order, same stack contents, hence the same frames: single-leaf mesh (nothing to fuse),
    one-level tree, depth-12 tree -- and a tree whose boxes are NOT the unions of their children's (one inner box inflated): rt_upload_bvh must refuse to fuse it
    (RT_SCENE_NOT_FUSED) and the option must fall back to the 64-byte records.  Besides frames: the reference GLSL's own traversal answers for the `crate`
    height field (552 rays with two or more triangles at the bit-equal winning t: the normal shows which one the visiting order left standing)."""
    monkeypatch.setenv("RT_FUSED", "1")
    W, H = 96, 64
    if mesh == "one_leaf":
        tris9 = np.array([[-1, 0, -1, 1, 0, -1, 0, 1.5, -1.2], [-1, 0, 1, 1, 0, 1, 0, 1.5, 0.5]], np.float32) + np.float32(0.25)
        nodes, tris = rt.build_bvh(tris9)
    else:
        nodes, tris = scenes.bunny_bvh(0 if mesh == "tiny" else 4)
    if inflate:
        nodes = nodes.copy()
        count = (nodes[:, 9] + 0.5).astype(int)
        k = int(np.nonzero(count == 0)[0][5])
        nodes[k, 0:3] -= np.float32(0.01)                # a legal, looser box: the reference (and the oracle) simply test it as it is
    faces = scenes.tiny_env(8)
    p = rt.default_render_params()
    p.sppPerFrame = 2
    cam = scenes.camera("closeup", aspect=W / H)
    with rt.Renderer(pipeline=rt.RT_PIPELINE_WAVEFRONT) as r:
        r.upload_bvh(nodes, tris)
        info = r.scene_info()
        if mesh == "one_leaf":
            assert info.nFused == 0 and not (info.flags & rt.RT_SCENE_NOT_FUSED)
        elif inflate:
            assert info.nFused == 0 and (info.flags & rt.RT_SCENE_NOT_FUSED)
        else:
            assert info.nFused > 0 and not (info.flags & rt.RT_SCENE_NOT_FUSED)
        r.upload_env(faces)
        r.resize(W, H)
        prev = None
        for f in range(2):
            u = rt.frame_uniforms(p, cam, W, H, f, True, nodes.shape[0], tris.shape[0])
            r.render_frame(u)
            want, _ = orc.render(u, nodes, tris, faces, prev)
            _assert_targets_equal(r.read_all(), want, orc, f"RT_FUSED=1 {mesh} inflate={inflate} frame={f}")
            prev = want[0]
        if mesh == "deep" and not inflate:
            from pathlib import Path
            d = np.load(Path(__file__).resolve().parent / "golden" / "glsl_bvh_trace_kat.npz")
            for tag in ("crate", "bunny"):
                kn, kt, rays = d[f"{tag}_nodes12"], d[f"{tag}_tris12"], d[f"{tag}_rays"]
                r.upload_bvh(kn, kt)
                assert r.scene_info().nFused > 0
                eps, inf = float(d["eps"]), float(d["inf"])
                mega = r.debug_trace(0, rays[:, 0:3], rays[:, 4:7], eps=eps, inf=inf)     # the walk tests/test_gpu_glsl_reference.py pins to the reference GLSL's answers
                fused = r.debug_trace(2, rays[:, 0:3], rays[:, 4:7], eps=eps, inf=inf)    # the production kernel on the fused records
                assert np.array_equal(fused[:, 0].view(np.uint32), mega[:, 0].view(np.uint32)), tag
                hit = mega[:, 0] < np.float32(inf)
                tri = fused[hit, 1].astype(int)
                nrm = np.cross(kt[tri, 4:7].astype(np.float64), kt[tri, 8:11].astype(np.float64))
                nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
                assert np.all(np.abs(np.sum(nrm * mega[hit, 4:7].astype(np.float64), axis=1) - 1.0) < 1e-5), tag   # the same winner on every tie
                assert hit.sum() >= 900


@pytest.fixture(scope="module")
def ren_wave():
    r = rt.Renderer(pipeline=rt.RT_PIPELINE_WAVEFRONT)
    yield r
    r.close()


@pytest.mark.parametrize("cam_kind,spp", [("closeup", 1), ("closeup", 3), ("default", 4)])
def test_bvh_scene_frames_wavefront(ren_wave, orc, cam_kind, spp):
    """The production pipeline (staged kernels, persistent traversal, ray queues) against the oracle."""
    W, H = 200, 120
    nodes, tris = scenes.bunny_bvh(4)
    faces = scenes.tiny_env(16)
    r = ren_wave
    r.upload_bvh(nodes, tris)
    r.upload_env(faces)
    r.resize(W, H)
    p = rt.default_render_params()
    p.sppPerFrame = spp
    cam = scenes.camera(cam_kind, aspect=W / H)
    prev = None
    for frame in range(3):
        u = rt.frame_uniforms(p, cam, W, H, frame, True, nodes.shape[0], tris.shape[0])
        r.render_frame(u)
        want, _ = orc.render(u, nodes, tris, faces, prev)
        _assert_targets_equal(r.read_all(), want, orc, f"wavefront {cam_kind} spp={spp} frame={frame}")
        prev = want[0]


@pytest.mark.parametrize("cap", [64, 1024])
def test_shadow_queue_2_overflow_is_traced_in_place(orc, monkeypatch, cap):
    """Round 5: a large launch set sizes shadow queue 2 (six ray records per bounce HIT) from the bounce hits of earlier batches instead of for the worst case, and
    the (hit, sample) pairs beyond that capacity trace their six rays in place (k_gen_gi_overflow, the megakernel's any-hit walk) instead of queueing them.  RT_Q2_CAP
    forces a capacity of 64 / 1024 pairs on a frame whose bounce rays hit the mesh thousands of times: the any-hit launch then traces only the queued part, the frames
    must still be the oracle's bit for bit -- frame by frame and as a batch of three."""
    W, H = 200, 120
    # the stand-in mesh twice, the second copy shifted so that the two face each other: bounce rays leaving one hit the other
    v, f = rt.meshgen.bunny_standin(4)
    a = rt.gather_triangles(v, f)
    b = a.copy()
    b[:, 0] += np.float32(0.7); b[:, 2] += np.float32(0.5)           # rows are (v0, e1, e2): only the base vertex moves
    nodes, tris = rt.build_bvh(np.concatenate([a, b]).astype(np.float32))
    faces = scenes.tiny_env(16)
    p = rt.default_render_params()
    p.sppPerFrame = 3
    cam = scenes.camera("closeup", aspect=W / H)
    us = [rt.frame_uniforms(p, cam, W, H, f_, True, nodes.shape[0], tris.shape[0]) for f_ in range(3)]
    monkeypatch.delenv("RT_Q2_CAP", raising=False)        # (the stress matrix sets it for the whole suite)
    with rt.Renderer(pipeline=rt.RT_PIPELINE_WAVEFRONT) as r:
        r.upload_bvh(nodes, tris); r.upload_env(faces); r.resize(W, H)
        for u in us:
            r.render_frame(u)
        full = r.traced_rays(True).shadow                 # (the one any-hit launch tallies both shadow queues)
    monkeypatch.setenv("RT_Q2_CAP", str(cap))
    with rt.Renderer(pipeline=rt.RT_PIPELINE_WAVEFRONT) as r, rt.Renderer(pipeline=rt.RT_PIPELINE_WAVEFRONT) as rb:
        r.upload_bvh(nodes, tris); r.upload_env(faces); r.resize(W, H)
        rb.upload_bvh(nodes, tris); rb.upload_env(faces); rb.resize(W, H)
        prev = None
        for f, u in enumerate(us):
            r.render_frame(u)
            want, _ = orc.render(u, nodes, tris, faces, prev)
            _assert_targets_equal(r.read_all(), want, orc, f"RT_Q2_CAP={cap} frame={f}")
            prev = want[0]
        capped = r.traced_rays(True).shadow
        rb.render_frames(us)
        _assert_targets_equal(rb.read_all(), want, orc, f"RT_Q2_CAP={cap} batch of three")
    assert full - capped > 2000, (full, capped)      # the queue really was too small: thousands of bounce-hit shadow rays were traced in place, not by the launch


@pytest.mark.parametrize("spp,W,H", [(16, 96, 64), (64, 48, 40)])
def test_bvh_wavefront_high_spp(ren_wave, orc, spp, W, H):
    """BASELINE configs 3-5 run at 16 and 64 spp: the ray queues hold 6*spp + aoSamples slots per hit, frames stay bit-exact."""
    nodes, tris = scenes.bunny_bvh(3)
    faces = scenes.tiny_env(8)
    r = ren_wave
    r.upload_bvh(nodes, tris)
    r.upload_env(faces)
    r.resize(W, H)
    p = rt.default_render_params()
    p.sppPerFrame = spp
    cam = scenes.camera("closeup", aspect=W / H)
    prev = None
    for frame in range(2):
        u = rt.frame_uniforms(p, cam, W, H, frame, True, nodes.shape[0], tris.shape[0])
        r.render_frame(u)
        want, cnt = orc.render(u, nodes, tris, faces, prev)
        _assert_targets_equal(r.read_all(), want, orc, f"wavefront spp={spp} frame={frame}")
        prev = want[0]
    assert cnt.raysClosest >= cnt.hitPixels * spp


@pytest.mark.parametrize("toggles", [dict(enableGI=0), dict(enableAO=0), dict(sunEnabled=0, pointLightEnabled=0),
                                     dict(enableEnvMap=0, skyEnabled=0), dict(enableTAA=0, enableJitter=0), dict(aoSamples=7)])
def test_bvh_wavefront_feature_toggles(ren_wave, orc, toggles):
    W, H = 96, 80
    nodes, tris = scenes.bunny_bvh(3)
    faces = scenes.tiny_env(8)
    r = ren_wave
    r.upload_bvh(nodes, tris)
    r.upload_env(faces)
    r.resize(W, H)
    p = rt.default_render_params()
    p.sppPerFrame = 2
    for k, v in toggles.items():
        setattr(p, k, v)
    cam = scenes.camera("closeup", aspect=W / H)
    prev = None
    for frame in range(2):
        u = rt.frame_uniforms(p, cam, W, H, frame, True, nodes.shape[0], tris.shape[0])
        r.render_frame(u)
        want, _ = orc.render(u, nodes, tris, faces, prev)
        _assert_targets_equal(r.read_all(), want, orc, f"toggles {toggles} frame={frame}")
        prev = want[0]


def test_full_size_wavefront_equals_megakernel(orc):
    """BASELINE configs[1] at full size (1080p, 4 spp, 81 920 triangles): too big for the oracle in a test,
    so use the size-independent property that two independent GPU formulations (megakernel, wavefront)
    of the same float model agree bit for bit, and that a run is reproducible; an oracle spot-check covers
    a 64x32 window of the same frame."""
    W, H = 1920, 1080
    nodes, tris = scenes.bunny_bvh(6)
    faces = scenes.env_faces("Sky_01")
    p = rt.default_render_params()
    p.sppPerFrame = 4
    cam = scenes.camera("closeup")
    outs = {}
    for name, pipe in (("mega", rt.RT_PIPELINE_MEGAKERNEL), ("wave", rt.RT_PIPELINE_WAVEFRONT), ("wave2", rt.RT_PIPELINE_WAVEFRONT)):
        with rt.Renderer(pipeline=pipe) as r:
            r.upload_bvh(nodes, tris)
            r.upload_env(faces)
            r.resize(W, H)
            for frame in range(2):
                r.render_frame(rt.frame_uniforms(p, cam, W, H, frame, True, nodes.shape[0], tris.shape[0]))
            outs[name] = r.read_all()
    for a, b in (("mega", "wave"), ("wave", "wave2")):
        for x, y in zip(outs[a], outs[b]):
            assert np.array_equal(x, y), (a, b)
    # oracle window on frame 0 (history-free): re-render frame 0 only
    with rt.Renderer() as r:
        r.upload_bvh(nodes, tris)
        r.upload_env(faces)
        r.resize(W, H)
        u = rt.frame_uniforms(p, cam, W, H, 0, True, nodes.shape[0], tris.shape[0])
        r.render_frame(u)
        got = r.read_all()
    x0, y0, x1, y1 = 900, 500, 964, 532
    want, _ = orc.render(u, nodes, tris, faces, None, region=(x0, y0, x1, y1))
    for g, w in zip(got, want):
        assert np.array_equal(g[y0:y1, x0:x1], w[y0:y1, x0:x1])


@pytest.mark.parametrize("cam_kind,spp", [("closeup", 1), ("closeup", 2), ("default", 4)])
def test_bvh_scene_frames(ren, orc, cam_kind, spp):
    W, H = 200, 120   # ragged: not a multiple of the 16-pixel tile
    nodes, tris = scenes.bunny_bvh(4)   # 5120 triangles
    faces = scenes.tiny_env(16)
    ren.upload_bvh(nodes, tris)
    ren.upload_env(faces)
    ren.resize(W, H)
    p = rt.default_render_params()
    p.sppPerFrame = spp
    cam = scenes.camera(cam_kind, aspect=W / H)
    prev = None
    for frame in range(3):
        u = rt.frame_uniforms(p, cam, W, H, frame, True, nodes.shape[0], tris.shape[0])
        ren.reset_counters()
        ren.render_frame(u)
        got = ren.read_all()
        want, cnt = orc.render(u, nodes, tris, faces, prev)
        _assert_targets_equal(got, want, orc, f"bvh {cam_kind} spp={spp} frame={frame}")
        c = ren.counters()
        assert (c.raysClosest, c.raysShadow, c.nodeFetch, c.triFetch, c.envLookup, c.hitPixels) == \
               (cnt.raysClosest, cnt.raysShadow, cnt.nodeFetch, cnt.triFetch, cnt.envLookup, cnt.hitPixels)
        assert (c.fetchPrimary, c.fetchShadow, c.fetchAO) == (cnt.fetchPrimary, cnt.fetchShadow, cnt.fetchAO)
        if cam_kind == "closeup":
            assert cnt.hitPixels > W * H // 5
        prev = want[0]


def test_render_ray_mirrors_mainloop(ren, orc):
    """rt_render_ray = mainLoop steps + renderRay (application.cpp:381-459): frame counter, jitter, history."""
    W, H = 96, 64
    ren.upload_env(None)
    ren.resize(W, H)
    p = rt.default_render_params()
    cam = scenes.camera("default", aspect=W / H)
    prev = None
    for frame in range(3):
        ren.render_ray(p, cam, use_bvh=False)
        u = orc.frame_uniforms(orc.default_render_params(), cam, W, H, frame, False)
        want, _ = orc.render(u, env_faces=np.array([[[[128, 128, 255, 255]]]] * 6, np.uint8), prev=prev)
        _assert_targets_equal(ren.read_all(), want, orc, f"render_ray frame={frame}")
        prev = want[0]
    assert ren.frame_index == 3
    ren.reset_accum()
    assert ren.frame_index == 0


def test_wavefront_chunked_queues_match_unchunked(orc):
    """Ray queues are processed in chunks of hits when the frame does not fit the queue budget; the chunked run
    (RT_QUEUE_BUDGET_MB=1 -> 4096-hit chunks) must be bit-identical to the oracle as well."""
    import os
    import subprocess
    import sys
    code = r'''
import sys, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import opengl_raytracing_amd as rt, oracle as orc, scenes
W, H = 200, 120
nodes, tris = scenes.bunny_bvh(4); faces = scenes.tiny_env(16)
p = rt.default_render_params(); p.sppPerFrame = 2
cam = scenes.camera("closeup", aspect=W / H)
with rt.Renderer(pipeline=rt.RT_PIPELINE_WAVEFRONT) as r:
    r.upload_bvh(nodes, tris); r.upload_env(faces); r.resize(W, H)
    prev = None
    for f in range(2):
        u = rt.frame_uniforms(p, cam, W, H, f, True, nodes.shape[0], tris.shape[0])
        r.render_frame(u)
        want, _ = orc.render(u, nodes, tris, faces, prev)
        for g, w in zip(r.read_all(), want):
            assert np.array_equal(g, w)
        prev = want[0]
print("CHUNKED-OK")
'''
    env = dict(os.environ, RT_QUEUE_BUDGET_MB="1")
    out = subprocess.run([sys.executable, "-c", code], cwd=str(scenes.ROOT), env=env, capture_output=True, text=True, timeout=300)
    assert "CHUNKED-OK" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


@pytest.mark.parametrize("lanes", [None, 1])
@pytest.mark.parametrize("pipeline", ["mega", "wave"])
@pytest.mark.parametrize("use_bvh", [False, True])
def test_moving_camera_reprojection(orc, monkeypatch, pipeline, use_bvh, lanes):
    """Camera path: rt_render_ray keeps FrameState (prev/curr view-projection), raises cameraMoved, switches the
    jitter scale (application.cpp:387-405); the shader writes motion vectors and resolveTAA takes the reprojection
    branch (rt_taa.glsl:116-179), with the (4,4) disocclusion marker on misses (rt.frag:172-175).
    lanes=1 (RT_LANES=1, one frame in flight): the history the reprojection reads must still be a buffer of its own -- with a
    COLOR0 ring of one a moving frame read texels its own resolve was overwriting (found in round 3)."""
    if pipeline == "wave" and not use_bvh:
        pytest.skip("the analytic scene always runs in the megakernel")
    if lanes:
        monkeypatch.setenv("RT_LANES", str(lanes))
    W, H = 160, 96
    nodes, tris = scenes.bunny_bvh(3)
    faces = scenes.tiny_env(8)
    p = rt.default_render_params()
    p.sppPerFrame = 2
    cam = scenes.camera("closeup" if use_bvh else "default", aspect=W / H)
    pipe = rt.RT_PIPELINE_MEGAKERNEL if pipeline == "mega" else rt.RT_PIPELINE_WAVEFRONT
    with rt.Renderer(pipeline=pipe) as r:
        r.upload_bvh(nodes, tris)
        r.upload_env(faces)
        r.resize(W, H)
        prev, prev_vp = None, None
        moved_frames = 0
        for frame in range(5):
            if frame in (1, 2, 4):          # move on some frames, hold still on others
                cam.pos[0] += 0.03
                cam.yaw += 0.4
            view, proj = orc.camera_view(cam), orc.camera_proj(cam)
            vp = orc.mat4_mul(proj, view)
            if prev_vp is None:
                prev_vp = vp
            moved = orc.camera_moved(vp, prev_vp)
            moved_frames += int(moved)
            u = orc.make_uniforms(orc.default_render_params() if False else p, cam, view, vp, prev_vp, W, H, frame, moved, use_bvh, False,
                                  nodes.shape[0], tris.shape[0], True)
            r.render_ray(p, cam, use_bvh=use_bvh)
            want, _ = orc.render(u, nodes, tris, faces, prev)
            _assert_targets_equal(r.read_all(), want, orc, f"moving {pipeline} bvh={use_bvh} frame={frame}")
            if moved:
                mot = orc.half_to_float(want[1])
                assert np.abs(mot).max() > 0       # real motion vectors were produced
            prev, prev_vp = want[0], vp
        assert moved_frames == 3


def test_million_triangle_scene_deep_tree(orc):
    """BASELINE configs[4]'s scene (16 objects, 1 M triangles, tree depth 18): exercises the 24-entry closest-hit and
    36-entry any-hit LDS stacks and the chunked queues.  Full-frame oracle runs are too slow for a test, so: wavefront ==
    megakernel bit for bit on a 640x360 frame, plus an oracle window."""
    v, f = rt.meshgen.million_triangle_scene()
    tris9 = rt.gather_triangles(v, f, np.eye(4, dtype=np.float32).reshape(-1))
    nodes, tris = rt.build_bvh(tris9)
    assert tris.shape[0] == 1_000_000
    faces = scenes.tiny_env(16)
    W, H = 640, 360
    p = rt.default_render_params()
    p.sppPerFrame = 2
    cam = scenes.camera("default", aspect=W / H)
    outs = {}
    for name, pipe in (("mega", rt.RT_PIPELINE_MEGAKERNEL), ("wave", rt.RT_PIPELINE_WAVEFRONT)):
        with rt.Renderer(pipeline=pipe) as r:
            r.upload_bvh(nodes, tris)
            r.upload_env(faces)
            r.resize(W, H)
            for frame in range(2):
                u = rt.frame_uniforms(p, cam, W, H, frame, True, nodes.shape[0], tris.shape[0])
                r.render_frame(u)
            outs[name] = r.read_all()
    for a, b in zip(outs["mega"], outs["wave"]):
        assert np.array_equal(a, b)
    assert outs["wave"][2].any()           # geometry was hit
    with rt.Renderer() as r:
        r.upload_bvh(nodes, tris)
        r.upload_env(faces)
        r.resize(W, H)
        u = rt.frame_uniforms(p, cam, W, H, 0, True, nodes.shape[0], tris.shape[0])
        r.render_frame(u)
        got = r.read_all()
    x0, y0, x1, y1 = 300, 100, 340, 120
    want, _ = orc.render(u, nodes, tris, faces, None, region=(x0, y0, x1, y1))
    for g, w_ in zip(got, want):
        assert np.array_equal(g[y0:y1, x0:x1], w_[y0:y1, x0:x1])

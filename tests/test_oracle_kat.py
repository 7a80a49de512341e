"""Pinning the oracle (CPU, no GPU): the known-answer vectors SURVEY.md 8c derives from the shader text,
structural KATs, GL-semantics checks, and the committed golden fixtures (tests/golden/).

The reference has no tests, goldens or fixtures for this path; besides the GLSL-executed fixtures of
tests/test_glsl_reference.py (analytic and BVH frames, traversal loops ray by ray) these pin the integer side exactly:
uint32 RNG KATs are exact; float KATs follow closed forms."""
import ctypes as C
import math
from pathlib import Path

import numpy as np
import pytest

import opengl_raytracing_amd as rt

GOLDEN = Path(__file__).resolve().parent / "golden"

RAND_KATS = [  # rand(p, frame) -> hash (uint32), float
    ((0.5, 0.5), 0, 4275526158, 0.99547350),
    ((0.5, 0.5), 1, 2625985622, 0.61140990),
    ((960.5, 540.5), 0, 443050774, 0.10315580),
    ((960.5, 540.5), 7, 2178151486, 0.50714040),
    ((1919.5, 1079.5), 123, 1921813567, 0.44745708),
    ((100.5, 200.5), 2047, 1328220868, 0.30925053),
]


def test_rand_kats(orc):
    L = orc.lib()
    for (px, py), frame, h, f in RAND_KATS:
        assert L.orc_rand_bits(px, py, frame) == h
        assert abs(L.orc_rand(px, py, frame) - f) < 5e-8
    # hash2 is pure uint32 arithmetic (rt_common.glsl:57-63): restate it here in Python ints
    def hash2(x, y):
        M = 0xFFFFFFFF
        x = (x * 1664525 + 1013904223) & M; y = (y * 1664525 + 1013904223) & M
        x ^= y >> 16; y ^= (x << 5) & M
        x = (x * 1664525 + 1013904223) & M; y = (y * 1664525 + 1013904223) & M
        return x ^ y
    rng = np.random.default_rng(0)
    for x, y in rng.integers(0, 2**32, size=(200, 2), dtype=np.uint64):
        assert L.orc_hash2(int(x), int(y)) == hash2(int(x), int(y))
    # frame * 1663 wraps in int32; float(uint) may round to exactly 1.0
    assert L.orc_rand_bits(3.5, 4.5, 2_000_000_000) == hash2(3 ^ 2_000_000_000, 4 ^ ((2_000_000_000 * 1663) & 0xFFFFFFFF))


def test_ld2_and_halton(orc):
    out = (C.c_float * 2)()
    want = [(0.5, 1 / 3), (0.25, 2 / 3), (0.75, 1 / 9), (0.125, 4 / 9)]
    for i, (a, b) in enumerate(want):
        orc.lib().orc_ld2(i, out)
        assert abs(out[0] - a) < 1e-7 and abs(out[1] - b) < 1e-7


def test_host_jitter_reproduces_reference_halton_bug(orc):
    # src/app/application.cpp:28-47: "f *= 0.5f" for base 3 too -> these six values
    want = [(0, 0), (-.25, .5), (.25, -.25), (-.375, .25), (.125, .75), (-.125, 0)]
    for i, w in enumerate(want):
        assert tuple(orc.generate_jitter(i)) == w
    assert tuple(orc.generate_jitter(1024 + 3)) == want[3]   # idx = frameIndex & 1023


def test_default_camera_uniforms(orc):
    p, cam = orc.default_render_params(), orc.default_camera()
    u = orc.frame_uniforms(p, cam, 1920, 1080, 0, False)
    np.testing.assert_allclose(list(u.camFwd), [0, -0.173648, -0.984808], atol=2e-6)
    np.testing.assert_allclose(list(u.camRight), [1, 0, 0], atol=1e-6)
    np.testing.assert_allclose(list(u.camUp), [0, 0.984808, -0.173648], atol=2e-6)
    assert abs(u.tanHalfFov - 0.577350) < 1e-6 and abs(u.aspect - 1920 / 1080) < 1e-7
    np.testing.assert_allclose(list(u.sunDir), [0.579228, -0.573576, 0.579228], atol=2e-6)
    np.testing.assert_allclose(list(u.skyUpDir), [0, 1, 0], atol=1e-6)
    assert (u.eps, u.inf) == (np.float32(1e-4), np.float32(1e30)) and abs(u.pi - 3.1415926535) < 1e-7
    assert u.cameraMoved == 0 and u.spp == 1 and u.useEnvMap == 1 and list(u.jitter) == [0.0, 0.0]
    m = orc.default_bvh_transform().reshape(4, 4).T
    np.testing.assert_array_equal(m, [[.5, 0, 0, -2], [0, .5, 0, 1.5], [0, 0, .5, 0], [0, 0, 0, 1]])
    # glm::perspective(60deg, 16/9, 0.1, 100) closed form
    P = orc.camera_proj(cam).reshape(4, 4).T
    t = math.tan(math.radians(60) / 2)
    np.testing.assert_allclose([P[0, 0], P[1, 1], P[2, 2], P[3, 2], P[2, 3]],
                               [1 / (16 / 9 * t), 1 / t, -(100.1) / 99.9, -1, -2 * 100 * 0.1 / 99.9], rtol=1e-6)


def test_oracle_math_close_to_libm(orc):
    L = orc.lib()

    def ulps(got, ref):
        ref32 = np.float32(ref)
        return abs(float(np.float32(got)) - ref) / max(float(np.spacing(abs(ref32))), 1e-45)
    xs = np.linspace(-7.0, 7.0, 4001, dtype=np.float32)
    assert max(ulps(L.orc_sin(float(x)), math.sin(float(x))) for x in xs if abs(math.sin(float(x))) > 1e-3) < 2.5
    assert max(ulps(L.orc_cos(float(x)), math.cos(float(x))) for x in xs if abs(math.cos(float(x))) > 1e-3) < 2.5
    ts = np.linspace(-30, 30, 2001, dtype=np.float32)
    assert max(ulps(L.orc_exp2(float(t)), 2.0 ** float(t)) for t in ts) < 2.0
    for x, y in [(0.9, 32.0), (0.5, 16.0), (0.99, 48.0), (0.2, 2.0), (0.3, 5.0), (1.0, 256.0)]:
        assert abs(L.orc_pow(x, y) - float(np.float32(x)) ** y) <= 3e-5 * float(np.float32(x)) ** y
    assert L.orc_pow(0.0, 32.0) == 0.0 and L.orc_pow(1.0, 32.0) == 1.0 and math.isnan(L.orc_pow(-1.0, 2.0))


def test_fp16_rounding_is_ieee_rne(orc):
    L = orc.lib()
    rng = np.random.default_rng(7)
    x = np.concatenate([rng.uniform(-70000, 70000, 2000), rng.uniform(-1e-4, 1e-4, 2000), rng.uniform(-2, 2, 2000),
                        [0.0, -0.0, 65504.0, 65519.99, 65520.0, 1e-8, 6.1e-5, 5.96e-8, 2.98e-8, 2.99e-8]]).astype(np.float32)
    with np.errstate(over="ignore"):
        want = x.astype(np.float16).view(np.uint16)
    got = np.array([L.orc_f32_to_f16(float(v)) for v in x], np.uint16)
    assert np.array_equal(got, want)
    h = np.arange(0, 0x7C00, 37, dtype=np.uint16)
    back = np.array([L.orc_f16_to_f32(int(v)) for v in h], np.float32)
    assert np.array_equal(back, h.view(np.float16).astype(np.float32))


def test_bvh_structural_kats(orc):
    rng = np.random.default_rng(3)
    for n in (1, 5, 8):   # <= 8 triangles: a single leaf {left=-1,right=-1,first=0,count=n}, src/scene/bvh.cpp:62-67
        t = rng.normal(size=(n, 9)).astype(np.float32)
        nodes, tris = orc.build_bvh(t)
        assert nodes.shape[0] == 1 and list(nodes[0, [3, 7, 8, 9]]) == [-1, -1, 0, n]
        assert np.array_equal(tris[:, [0, 1, 2, 4, 5, 6, 8, 9, 10]], t)
    t = rng.normal(size=(100, 9)).astype(np.float32)
    nodes, tris = orc.build_bvh(t)
    assert nodes[0, 3] == 1                       # pre-order: node 1 is the root's left child
    assert nodes[0, 9] == 0 and nodes[0, 7] > 1
    leaves = nodes[nodes[:, 9] > 0]
    assert leaves[:, 9].sum() == 100 and leaves[:, 9].max() <= 8
    # DFS re-packing visits the RIGHT subtree first (LIFO stack, bvh.cpp:130-131): the right-most leaf has first == 0
    n = 0
    while nodes[n, 9] == 0:
        n = int(nodes[n, 7])
    assert nodes[n, 8] == 0
    assert sorted(map(tuple, tris[:, [0, 1, 2, 4, 5, 6, 8, 9, 10]])) == sorted(map(tuple, t))   # a permutation of the input
    # every node's box bounds its triangles
    def tri_pts(i):
        v0, e1, e2 = tris[i, 0:3], tris[i, 4:7], tris[i, 8:11]
        return np.stack([v0, v0 + e1, v0 + e2])
    for nd in leaves:
        pts = np.concatenate([tri_pts(i) for i in range(int(nd[8]), int(nd[8] + nd[9]))])
        assert np.array_equal(pts.min(0), nd[0:3]) and np.array_equal(pts.max(0), nd[4:7])
    assert orc.build_bvh(np.zeros((0, 9), np.float32))[0].shape[0] == 0


def test_cubemap_face_selection_and_filtering(orc):
    n = 4
    faces = np.zeros((6, n, n, 3), np.uint8)
    for f in range(6):
        faces[f, :, :, 0] = 40 * (f + 1)
        faces[f, :, :, 1] = (np.arange(n) * 50)[None, :]      # varies with s (column)
        faces[f, :, :, 2] = (np.arange(n) * 60)[:, None]      # varies with t (row)
    L = orc.lib()
    out = np.zeros(3, np.float32)

    def tex(d):
        d = np.asarray(d, np.float32)
        L.orc_texture_cube(faces.ctypes.data_as(C.POINTER(C.c_uint8)), n, 3, d.ctypes.data_as(C.POINTER(C.c_float)), out.ctypes.data_as(C.POINTER(C.c_float)))
        return out.copy()
    # major axis -> face (+X -X +Y -Y +Z -Z), OpenGL 4.1 core table 3.x
    for f, d in enumerate([(1, .1, .2), (-1, .1, .2), (.1, 1, .2), (.1, -1, .2), (.1, .2, 1), (.1, .2, -1)]):
        assert abs(tex(d)[0] - 40 * (f + 1) / 255) < 1e-6
    # +X face: sc = -z, tc = -y.  Towards -z  -> larger s -> larger column value
    assert tex((1, 0, -0.5))[1] > tex((1, 0, 0.5))[1]
    assert tex((1, -0.5, 0))[2] > tex((1, 0.5, 0))[2]
    # +Y face: sc = +x, tc = +z
    assert tex((0.5, 1, 0))[1] > tex((-0.5, 1, 0))[1] and tex((0, 1, 0.5))[2] > tex((0, 1, -0.5))[2]
    # texel centre sampling is exact; CLAMP_TO_EDGE at the face border (no seamless filtering)
    s = (1 + 0.5) / n * 2 - 1            # column 1 centre on +Z (sc = +x)
    t = (2 + 0.5) / n * 2 - 1            # row 2 centre (tc = -y)
    np.testing.assert_allclose(tex((s, -t, 1)), [200 / 255, 50 / 255, 120 / 255], atol=1e-6)
    np.testing.assert_allclose(tex((0.999999, 0, 1))[1], 150 / 255, atol=1e-6)
    # halfway between two texel centres -> mean of the two
    s = (1 + 1.0) / n * 2 - 1
    np.testing.assert_allclose(tex((s, -t, 1))[1], 75 / 255, atol=1e-6)


def test_cubemap_quantised_filter_mode(orc):
    """SURVEY.md 8c's second filter model (RtExtension.envFilter = 1): texel coordinates rounded to 1/256 texel before the bilinear weights.
    Same texels / faces / clamping; on face interiors the two modes differ by at most 1/255 per channel (coordinates move by <= 1/512 texel,
    a texel step is <= 255/255); at texel centres and at multiples of 1/256 they agree exactly."""
    rng = np.random.default_rng(5)
    n = 8
    faces = rng.integers(0, 256, (6, n, n, 3), dtype=np.uint8)
    L = orc.lib()
    out = np.zeros(3, np.float32)

    def tex(d, mode):
        d = np.asarray(d, np.float32)
        L.orc_set_env_filter(mode)
        L.orc_texture_cube(faces.ctypes.data_as(C.POINTER(C.c_uint8)), n, 3, d.ctypes.data_as(C.POINTER(C.c_float)), out.ctypes.data_as(C.POINTER(C.c_float)))
        L.orc_set_env_filter(0)
        return out.copy()
    worst, differ = 0.0, 0
    for d in rng.normal(size=(4000, 3)):
        a, b = tex(d, 0), tex(d, 1)
        worst = max(worst, float(np.abs(a - b).max()))
        differ += int(not np.array_equal(a, b))
    assert worst <= 1.0 / 255.0 + 1e-6 and differ > 3000          # a different model, within one 8-bit step
    # a coordinate on the 1/256 grid: +Z face, s*N - 0.5 = 2 + 64/256 exactly, t at a texel centre -> identical weights in both modes
    s_ = (2.25 + 0.5) / n * 2 - 1
    t_ = (3 + 0.5) / n * 2 - 1
    assert np.array_equal(tex((s_, -t_, 1), 0), tex((s_, -t_, 1), 1))
    # mode 1 snaps: coordinates 1/1024 texel apart give the same result
    eps_ = (1.0 / 1024) / n * 2
    assert np.array_equal(tex((s_ + eps_, -t_, 1), 1), tex((s_, -t_, 1), 1)) and not np.array_equal(tex((s_ + eps_, -t_, 1), 0), tex((s_, -t_, 1), 0))


def test_cubemap_cross_slicing(orc):
    n = 3
    img = np.zeros((3 * n, 4 * n, 3), np.uint8)
    cells = {0: (2, 1), 1: (0, 1), 2: (1, 0), 3: (1, 2), 4: (1, 1), 5: (3, 1)}   # src/render/cubemap.cpp:86-91
    for f, (cx, cy) in cells.items():
        img[cy * n:(cy + 1) * n, cx * n:(cx + 1) * n] = np.arange(n * n * 3, dtype=np.uint8).reshape(n, n, 3) + 20 * f
    faces = orc.cubemap_from_cross(img)
    for f in range(6):
        assert np.array_equal(faces[f], np.arange(n * n * 3, dtype=np.uint8).reshape(n, n, 3) + 20 * f)
    with pytest.raises(ValueError):
        orc.cubemap_from_cross(np.zeros((9, 10, 3), np.uint8))


def test_analytic_scene_structure(orc):
    """Centre pixel of the default camera looks at the floor; the TAA still-branch is the 0.85/0.15 EMA."""
    p = orc.default_render_params()
    p.enableEnvMap = 0
    p.enableJitter = 0
    cam = orc.default_camera()
    cam.aspect = 1.0
    W = H = 33
    u0 = orc.frame_uniforms(p, cam, W, H, 0, False)
    out0, c0 = orc.render(u0, region=(16, 16, 17, 17))
    gpos = orc.half_to_float(out0[2][16, 16])
    gnrm = orc.half_to_float(out0[3][16, 16])
    assert abs(gpos[1]) < 1e-3 and gpos[3] == 1.0 and list(gnrm[:3]) == [0, 1, 0]   # floor hit: y = 0, n = +Y
    assert list(orc.half_to_float(out0[1][16, 16])) == [0, 0]                      # static camera: zero motion
    # frame 1 = prev * 0.85 + curr * 0.15 (rt_taa.glsl:91-104), then fp16
    u1 = orc.frame_uniforms(p, cam, W, H, 1, False)
    p2 = p.copy(); p2.enableTAA = 0
    curr1, _ = orc.render(orc.frame_uniforms(p2, cam, W, H, 1, False), region=(16, 16, 17, 17))
    out1, _ = orc.render(u1, prev=out0[0], region=(16, 16, 17, 17))
    prev = orc.half_to_float(out0[0][16, 16]).astype(np.float32)
    # curr1 (TAA off) is fp16-rounded; recompute the blend in fp32 from the unrounded current colour is not possible here,
    # so check against the rounded one with one half-ulp of slack
    cur = orc.half_to_float(curr1[0][16, 16]).astype(np.float32)
    blend = prev * np.float32(0.85) + cur * np.float32(0.15)
    got = orc.half_to_float(out1[0][16, 16])
    np.testing.assert_allclose(got[:3], blend[:3], rtol=2e-3)
    # sky pixel (top row): gbuffer stays zero, one ray, no hit
    outs, cs = orc.render(u0, region=(16, 32, 17, 33))
    assert cs.hitPixels == 0 and cs.raysAnalytic == 1 and not outs[2][32, 16].any()


@pytest.mark.parametrize("name", ["analytic_gradient_64", "analytic_env_48", "bvh_closeup_48x32"])
def test_golden_fixtures(orc, name):
    d = np.load(GOLDEN / f"{name}.npz")
    nodes = d["nodes12"] if "nodes12" in d else None
    tris = d["tris12"] if "tris12" in d else None
    env = d["env"] if "env" in d else None
    prev = None
    for f in range(d["uniforms"].shape[0]):
        u = rt.RtUniforms.from_buffer_copy(d["uniforms"][f].tobytes())
        outs, cnt = orc.render(u, nodes, tris, env, prev)
        for k, a in zip(("color", "motion", "gpos", "gnrm"), outs):
            assert np.array_equal(a, d[f"{k}{f}"]), (name, f, k)
        gold = tuple(int(v) for v in d["counters"][f])   # fixtures hold the 7 base counters
        assert gold == cnt.as_tuple()[:len(gold)]
        assert cnt.fetchPrimary + cnt.fetchShadow + cnt.fetchAO <= cnt.nodeFetch + cnt.triFetch
        prev = outs[0]


def test_region_and_mask_rendering_is_consistent(orc):
    d = np.load(GOLDEN / "bvh_closeup_48x32.npz")
    u = rt.RtUniforms.from_buffer_copy(d["uniforms"][0].tobytes())
    full, _ = orc.render(u, d["nodes12"], d["tris12"], d["env"], None)
    mask = np.zeros((32, 48), np.uint8)
    mask[::3, 1::2] = 1
    part, _ = orc.render(u, d["nodes12"], d["tris12"], d["env"], None, mask=mask, nthreads=3)
    m = mask.astype(bool)
    assert np.array_equal(part[0][m], full[0][m]) and not part[0][~m].any()


def test_traversal_equals_brute_force_over_the_pinned_primitives(orc):
    """traceBVH / traceBVHShadow (rt_bvh.glsl:193-304) are the one part of the path the reference's GLSL could not be
    executed for (tests/test_glsl_reference.py); their primitives aabbHit / triHit are pinned.  The restated loops must
    then return what an exhaustive sweep with the pinned triHit returns: closest t bit-identical, any-hit identical."""
    import opengl_raytracing_amd as rt
    v, f = rt.meshgen.bunny_standin(2)
    nodes, tris = orc.build_bvh(orc.gather_triangles(v, f))
    n = tris.shape[0]
    u = orc.frame_uniforms(orc.default_render_params(), orc.default_camera(), 8, 8, 0, True, nodes.shape[0], n)
    rng = np.random.default_rng(17)
    centre = 0.5 * (nodes[0, 0:3] + nodes[0, 4:7])
    hits = shadowed = 0
    for k in range(160):
        ro = (centre + rng.normal(0, 1.0, 3) * 2.0).astype(np.float32)
        target = tris[rng.integers(n), 0:3] + rng.normal(0, 0.05, 3)
        rd = (target - ro) / np.linalg.norm(target - ro)
        rd = rd.astype(np.float32)
        tmax_shadow = np.float32(rng.uniform(0.5, 6.0))
        best, best_n, any_hit = np.float32(u.inf), None, False
        for i in range(n):
            r = orc.tri_hit(u, ro, rd, tris[i], best)          # closest: tMax = best so far, ties overwrite (rt_bvh.glsl:166,218)
            if r[0]:
                best, best_n = r[1], r[2:5].copy()
            any_hit = any_hit or bool(orc.tri_hit(u, ro, rd, tris[i], tmax_shadow)[0])
        hit, t, p, nn, _ = orc.trace_bvh(u, nodes, tris, ro, rd)
        assert hit == (best_n is not None), k
        if hit:
            hits += 1
            assert np.float32(t) == best, (k, t, best)
        assert orc.trace_bvh_shadow(u, nodes, tris, ro, rd, float(tmax_shadow)) == any_hit, k
        shadowed += any_hit
    assert hits > 60 and 20 < shadowed < 160


def test_native_baseline_build_is_bit_identical(orc, tmp_path):
    """bench.py's cpu_baseline times the oracle compiled `-O3 -march=native`.  Same sources, same float model
    (no contraction, fmaf = hardware FMA): every bit of a BVH frame and of an analytic frame, and every work counter, must
    equal the `-O2 -march=x86-64-v3` checker build's."""
    import opengl_raytracing_amd as rt
    L = orc.load(orc.build_native(tmp_path))
    v, f = rt.meshgen.bunny_standin(3)
    nodes, tris = orc.build_bvh(orc.gather_triangles(v, f))
    rng = np.random.default_rng(4)
    faces = rng.integers(0, 256, size=(6, 8, 8, 3), dtype=np.uint8)
    W, H = 96, 64
    p = orc.default_render_params()
    p.sppPerFrame = 2
    cam = orc.default_camera()
    cam.pos[0], cam.pos[1], cam.pos[2], cam.yaw, cam.pitch, cam.aspect = -2.0, 1.5, 1.0, -90.0, 0.0, W / H
    for use_bvh in (True, False):
        prev = None
        for frame in range(2):
            u = orc.frame_uniforms(p, cam, W, H, frame, use_bvh, nodes.shape[0], tris.shape[0])
            a, ca = orc.render(u, nodes, tris, faces, prev)
            b, cb = orc.render(u, nodes, tris, faces, prev, L=L)
            for x, y in zip(a, b):
                assert np.array_equal(x, y)
            assert ca.as_tuple() == cb.as_tuple() and ca.rays > 0
            prev = a[0]

"""The oracle against the REFERENCE's own GLSL, executed in the build container by SwiftShader's software
OpenGL ES 3.0 (oracle/glsl_ref.py, fixtures written by tests/golden/make_glsl_golden.py).

Each fixture holds complete inputs (uniform bytes, cube map, history) and the shader's outputs.  SwiftShader's
sin/cos/pow/normalize/dot differ from the oracle's float model in the last bits (no FMA, its own polynomials), so
agreement is to a tolerance: RMSE < 1e-4 per target (north_star's float tolerance) AND at least 99 % of all values
bit-identical -- i.e. what differs are isolated pixels next to discontinuities, not a systematic error.
"""
from pathlib import Path

import numpy as np
import pytest

import opengl_raytracing_amd as rt

GOLDEN = Path(__file__).resolve().parent / "golden"
FRAME_FIXTURES = ["glsl_analytic_gradient_64x48", "glsl_analytic_materials_env_48x36", "glsl_analytic_moving_48x36",
                  "glsl_analytic_toggles_48x36",
                  # rt.frag in BVH mode -- traceBVH / traceBVHShadow executed by the reference GLSL for every primary, shadow,
                  # bounce and AO ray of the frame (round 2: oracle/glsl_ref.py structured_continue)
                  "glsl_bvh_closeup_64x48", "glsl_bvh_moving_64x48", "glsl_bvh_5k_gradient_64x48"]
RMSE_TOL = 1e-4
EXACT_MIN = 0.99


def h2f(a):
    return a.view(np.float16).astype(np.float32)


def check_targets(name, f, got, d):
    for k, a in zip(("color", "motion", "gpos", "gnrm"), got):
        b = d[f"{k}{f}"]
        fa, fb = h2f(a), h2f(b)
        ok = np.isfinite(fa) & np.isfinite(fb)
        diff = np.where(ok, fa - fb, 0.0)
        rmse = float(np.sqrt(np.mean(diff * diff)))
        exact = float(np.mean(a == b))
        assert rmse < RMSE_TOL and exact >= EXACT_MIN, (name, f, k, rmse, exact)


@pytest.mark.parametrize("name", FRAME_FIXTURES)
def test_oracle_matches_reference_glsl_frames(orc, name):
    d = np.load(GOLDEN / f"{name}.npz")
    env = d["env"] if "env" in d else None
    prev = None
    for f in range(d["uniforms"].shape[0]):
        u = rt.RtUniforms.from_buffer_copy(d["uniforms"][f].tobytes())
        got, _ = orc.render(u, d["nodes12"] if "nodes12" in d else None, d["tris12"] if "tris12" in d else None, env, prev)
        check_targets(name, f, got, d)
        prev = d[f"color{f}"]          # the shader's own history, as in the fixture


def test_moving_fixture_exercises_reprojection():
    d = np.load(GOLDEN / "glsl_analytic_moving_48x36.npz")
    u = rt.RtUniforms.from_buffer_copy(d["uniforms"][1].tobytes())
    assert u.cameraMoved == 1
    m = h2f(d["motion1"])
    assert np.count_nonzero(m) > m.size // 4          # real motion vectors
    assert (m == 4.0).any()                           # and the disocclusion / sky marker of rt.frag


@pytest.mark.parametrize("tag", ["svgf", "plain", "motion", "svgf_moving"])
def test_oracle_matches_reference_glsl_present(orc, tag):
    d = np.load(GOLDEN / "glsl_present_48x36.npz")
    pp = rt.RtPresentParams.from_buffer_copy(d[f"pp_{tag}"].tobytes())
    targets = [d[f"{k}_{tag}"] for k in ("color", "motion", "gpos", "gnrm")]
    got = orc.present(pp, targets)
    want = d[f"rgba_{tag}"]
    diff = np.abs(got.astype(np.int32) - want.astype(np.int32))
    assert diff.max() <= 1 and np.mean(got == want) >= 0.995, (tag, diff.max(), np.mean(got == want))


def test_oracle_matches_reference_glsl_bvh_primitives(orc):
    """nodeFetch / triFetch / aabbHit / triHit of rt_bvh.glsl, one call per case (the loops: ..._traversal_loops below).  Slab tests must agree exactly; Moller-Trumbore t to a median relative error < 1e-6 (the oracle's
    dot/cross use fmaf, SwiftShader's do not) with the hit decision allowed to differ only on constructed borderline cases."""
    d = np.load(GOLDEN / "glsl_bvh_kat.npz")
    nodes, tris, rays, o0, o1, o2 = (d[k] for k in ("nodes12", "tris12", "rays", "o0", "o1", "o2"))
    n = rays.shape[0]
    u = orc.frame_uniforms(orc.default_render_params(), orc.default_camera(), 8, 8, 0, True, n, n)
    assert u.eps == d["eps"]
    # node decode (rt_bvh.glsl:97-100): left, right, first*8+count as the shader saw them
    left = np.trunc(nodes[:, 3] + 0.5).astype(np.int64)
    right = np.trunc(nodes[:, 7] + 0.5).astype(np.int64)
    first, count = np.trunc(nodes[:, 8] + 0.5).astype(np.int64), np.trunc(nodes[:, 9] + 0.5).astype(np.int64)
    assert np.array_equal(o0[:, 3].astype(np.int64), left) and np.array_equal(o1[:, 2].astype(np.int64), right)
    assert np.array_equal(o1[:, 3].astype(np.int64), first * 8 + count)
    flag_diff, hits, rel = 0, 0, []
    for i in range(n):
        a = orc.aabb_hit(rays[i, 0:3], rays[i, 4:7], nodes[i, 0:3], nodes[i, 4:7])
        assert a[0] == o0[i, 0], i
        assert (a[1] == o0[i, 1] or (np.isnan(a[1]) and np.isnan(o0[i, 1]))) and (a[2] == o0[i, 2] or (np.isnan(a[2]) and np.isnan(o0[i, 2]))), i
        t = orc.tri_hit(u, rays[i, 0:3], rays[i, 4:7], tris[i], rays[i, 3])
        if t[0] != o1[i, 0]:
            flag_diff += 1
            continue
        if t[0]:
            hits += 1
            rel.append(abs(t[1] - o1[i, 1]) / abs(o1[i, 1]))
            assert rel[-1] <= 5e-5, (i, t[1], o1[i, 1])             # grazing triangles (tiny det) amplify the last-bit differences
            assert np.max(np.abs(t[2:5] - o2[i, 0:3])) <= 1e-6, i
    assert hits > 100 and flag_diff <= n // 50, (hits, flag_diff)
    assert np.median(rel) < 1e-6 and np.percentile(rel, 95) < 5e-6


def check_taa_regimes(d, render):
    """render(u, prev) -> four targets.  Frames around the history-weight switches of resolveTAA (rt_taa.glsl:91-104: 0.85 below
    frame 8, 0.92 below 32, 0.96 from 32 on), each with the history the reference GLSL itself read."""
    for f in [int(x) for x in d["frames"]]:
        u = rt.RtUniforms.from_buffer_copy(d[f"uniforms{f}"].tobytes())
        assert u.frameIndex == f and u.cameraMoved == 0 and u.enableTAA == 1
        prev = d[f"prev{f}"] if f > 0 else None
        got = render(u, prev)
        for k, a in zip(("color", "motion", "gpos", "gnrm"), got):
            b = d[f"{k}{f}"]
            diff = np.where(np.isfinite(h2f(a)) & np.isfinite(h2f(b)), h2f(a) - h2f(b), 0.0)
            assert float(np.sqrt(np.mean(diff * diff))) < RMSE_TOL and float(np.mean(a == b)) >= EXACT_MIN, (f, k)


def test_oracle_matches_reference_glsl_taa_weight_regimes(orc):
    d = np.load(GOLDEN / "glsl_bvh_taa_regimes_48x36.npz")
    check_taa_regimes(d, lambda u, prev: orc.render(u, d["nodes12"], d["tris12"], d["env"], prev)[0])
    # the three weights really are in play: frame f blends (1 - w) of its own radiance into the history
    c7, c8 = h2f(d["color7"]), h2f(d["color8"])
    assert np.abs(c8 - c7).mean() > 0


def check_trace_kat(tag, d, closest, any_hit):
    """closest(i) -> (hit, t, p[3], n[3]); any_hit(i) -> bool, for ray i of fixture section `tag`.  Shared with the GPU test.
    Rays lying IN a box plane of an axis they do not move along make a slab product 0 * inf = NaN; GLSL leaves min/max of a NaN
    to the driver (SURVEY.md 8a aabbHit), so those rays are recorded in the fixture but not compared."""
    rays, o0, o1, o2, nan_slab, ties = (d[f"{tag}_{k}"] for k in ("rays", "o0", "o1", "o2", "nan_slab", "ties"))
    exact_arith = tag == "crate"              # integer coordinates, axis rays: every product is exact, so t must be bit-identical
    compared = hits = tie_rays = 0
    for i in range(rays.shape[0]):
        if nan_slab[i]:
            continue
        compared += 1
        hit, t, p, n = closest(i)
        assert bool(o0[i, 0]) == bool(hit), (tag, i, "hit")
        assert bool(o0[i, 2]) == bool(any_hit(i)), (tag, i, "shadow")
        if not hit:
            assert o0[i, 1] == d["inf"] and o0[i, 3] == 1.0          # hitOut.t = uINF, mat = 1 are set before the loop (rt_bvh.glsl:195-197)
            continue
        hits += 1
        tie_rays += ties[i] > 1
        if exact_arith:
            assert np.float32(t) == o0[i, 1], (tag, i, t, o0[i, 1])
        else:
            assert abs(t - o0[i, 1]) <= 5e-6 * abs(o0[i, 1]), (tag, i, t, o0[i, 1])
        assert np.max(np.abs(np.asarray(n) - o2[i, 0:3])) <= 1e-5, (tag, i, n, o2[i])      # which triangle won, incl. equal-t ties
        assert np.max(np.abs(np.asarray(p) - o1[i, 0:3])) <= 1e-5 * max(1.0, float(np.abs(o1[i, 0:3]).max())), (tag, i)
        assert o0[i, 3] == 1.0
    return compared, hits, tie_rays


@pytest.mark.parametrize("tag", ["crate", "bunny"])
def test_oracle_matches_reference_glsl_traversal_loops(orc, tag):
    """traceBVH (rt_bvh.glsl:193-243) and traceBVHShadow (:260-304) executed ray by ray by the reference GLSL
    (tests/golden/make_glsl_golden.py section H).  `crate`: exact-arithmetic height field, axis rays through shared edges:
    hundreds of rays where two triangles have bit-equal t, so the returned normal shows the visit order, the pop-time cull and
    the tie rule; rdInv = +-inf on two axes.  `bunny`: generic mesh, generic rays."""
    d = np.load(GOLDEN / "glsl_bvh_trace_kat.npz")
    nodes, tris, rays = d[f"{tag}_nodes12"], d[f"{tag}_tris12"], d[f"{tag}_rays"]
    u = orc.frame_uniforms(orc.default_render_params(), orc.default_camera(), 8, 8, 0, True, nodes.shape[0], tris.shape[0])
    assert u.eps == d["eps"] and u.inf == d["inf"]

    def closest(i):
        hit, t, p, n, _ = orc.trace_bvh(u, nodes, tris, rays[i, 0:3], rays[i, 4:7])
        return hit, t, p, n

    compared, hits, tie_rays = check_trace_kat(tag, d, closest, lambda i: orc.trace_bvh_shadow(u, nodes, tris, rays[i, 0:3], rays[i, 4:7], float(rays[i, 3])))
    assert compared >= 1000 and hits >= 900
    if tag == "crate":
        assert tie_rays >= 300, tie_rays


def test_structured_continue_rewrite():
    """The one load-time change to the traversal functions' text (oracle/glsl_ref.py): `if (C) continue;` followed by the rest
    of the loop body becomes `if (!(C)) {` rest `}` -- checked here on a stand-alone snippet (the reference is not readable
    from the test suite)."""
    import sys
    sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "oracle"))
    import glsl_ref
    src = "void f() {\n    while (sp > 0) {\n        int ni = stack[--sp];\n        if (!hit(ni) || t > best) continue;\n\n        if (leaf) {\n            g();   // }\n        } else {\n            h();\n        }\n    }\n    return;\n}"
    out, n = glsl_ref.structured_continue(src)
    assert n == 1
    assert out.split("\n") == ["void f() {", "    while (sp > 0) {", "        int ni = stack[--sp];", "        if (!(!hit(ni) || t > best)) {", "",
                               "        if (leaf) {", "            g();   // }", "        } else {", "            h();", "        }", "        }", "    }",
                               "    return;", "}"]
    assert glsl_ref.structured_continue("x = 1;\ncontinue_label();")[1] == 0


@pytest.mark.parametrize("tag", ["default", "disk_light_only", "no_env_gi_only"])
def test_oracle_matches_reference_glsl_bvh_shading(orc, tag):
    """rt.frag:92-106 (directLightBVH + giScaleBVH * oneBounceGIBVH, times computeAO) for 512 given hits with an empty BVH,
    executed by the reference GLSL (fp32 outputs, no fp16 rounding to hide behind): relative error < 5e-5, median < 1e-6."""
    d = np.load(GOLDEN / "glsl_bvh_shade_kat.npz")
    u = rt.RtUniforms.from_buffer_copy(d[f"u_{tag}"].tobytes())
    assert u.useBVH == 1 and u.nodeCount == 0
    got = orc.shade_bvh_hits(u, d["env"] if u.useEnvMap == 1 else None, d["hits"])
    want = d[f"rad_{tag}"]
    rel = np.abs(got - want) / np.maximum(np.abs(want), 1e-3)
    assert rel.max() < 5e-5 and np.median(rel) < 1e-6, (tag, rel.max(), np.median(rel))
    assert want.mean() > 0.05 and np.count_nonzero(want) > want.size // 2

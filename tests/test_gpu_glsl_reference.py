"""The HIP path (through the C ABI) against the REFERENCE's own GLSL outputs (tests/golden/glsl_*.npz, produced by
tests/golden/make_glsl_golden.py: the reference shaders executed by SwiftShader).  Same inputs, same history (handed over
with rt_write_target), same tolerances as the oracle's check in tests/test_glsl_reference.py -- the product is compared
with the reference directly, not only through the oracle."""
from pathlib import Path

import numpy as np
import pytest

import opengl_raytracing_amd as rt
from test_glsl_reference import FRAME_FIXTURES, check_targets, check_taa_regimes, check_trace_kat

pytestmark = pytest.mark.gpu
GOLDEN = Path(__file__).resolve().parent / "golden"


@pytest.fixture(scope="module")
def ren():
    r = rt.Renderer()
    yield r
    r.close()


@pytest.mark.parametrize("name", FRAME_FIXTURES)
def test_hip_matches_reference_glsl_frames(ren, name):
    d = np.load(GOLDEN / f"{name}.npz")
    u0 = rt.RtUniforms.from_buffer_copy(d["uniforms"][0].tobytes())
    ren.resize(int(u0.resolution[0]), int(u0.resolution[1]))
    ren.upload_env(d["env"] if "env" in d else None)
    if "nodes12" in d:
        ren.upload_bvh(d["nodes12"], d["tris12"])
    ren.reset_accum()
    for f in range(d["uniforms"].shape[0]):
        u = rt.RtUniforms.from_buffer_copy(d["uniforms"][f].tobytes())
        if f > 0:
            ren.write_target(rt.RT_TARGET_COLOR, d[f"color{f - 1}"])      # the shader's own history, as in the fixture
        ren.render_frame(u)
        check_targets(name, f, ren.read_all(), d)


@pytest.mark.parametrize("name", [n for n in FRAME_FIXTURES if "_bvh_" in n])
def test_hip_megakernel_matches_reference_glsl_bvh_frames(name):
    """The default renderer takes the wavefront pipeline for BVH frames (test above); the same fixtures through the megakernel."""
    d = np.load(GOLDEN / f"{name}.npz")
    u0 = rt.RtUniforms.from_buffer_copy(d["uniforms"][0].tobytes())
    with rt.Renderer(pipeline=rt.RT_PIPELINE_MEGAKERNEL) as r:
        r.resize(int(u0.resolution[0]), int(u0.resolution[1]))
        r.upload_env(d["env"] if "env" in d else None)
        r.upload_bvh(d["nodes12"], d["tris12"])
        for f in range(d["uniforms"].shape[0]):
            if f > 0:
                r.write_target(rt.RT_TARGET_COLOR, d[f"color{f - 1}"])
            r.render_frame(rt.RtUniforms.from_buffer_copy(d["uniforms"][f].tobytes()))
            check_targets(name, f, r.read_all(), d)


@pytest.mark.parametrize("pipeline", [rt.RT_PIPELINE_WAVEFRONT, rt.RT_PIPELINE_MEGAKERNEL])
def test_hip_matches_reference_glsl_taa_weight_regimes(pipeline):
    """Frames 0, 1, 7, 8, 9, 31, 32, 33 of a 34-frame accumulation as the reference GLSL rendered them; the library takes uFrameIndex
    from its own frame counter, so the frames in between are rendered too (their output is irrelevant: every checked frame gets
    the reference's own history through rt_write_target)."""
    d = np.load(GOLDEN / "glsl_bvh_taa_regimes_48x36.npz")
    u0 = rt.RtUniforms.from_buffer_copy(d["uniforms0"].tobytes())
    with rt.Renderer(pipeline=pipeline) as r:
        r.resize(int(u0.resolution[0]), int(u0.resolution[1]))
        r.upload_env(d["env"])
        r.upload_bvh(d["nodes12"], d["tris12"])

        def render(u, prev):
            while r.frame_index < u.frameIndex:           # advance the frame counter to the fixture's frame
                r.render_frame(u)
            assert r.frame_index == u.frameIndex
            if prev is not None:
                r.write_target(rt.RT_TARGET_COLOR, prev)
            r.render_frame(u)
            return r.read_all()

        check_taa_regimes(d, render)


@pytest.mark.parametrize("tag", ["crate", "bunny"])
def test_hip_matches_reference_glsl_traversal_loops(ren, tag):
    """rt_debug_trace (the device traversal) against traceBVH / traceBVHShadow as the reference GLSL executed them."""
    d = np.load(GOLDEN / "glsl_bvh_trace_kat.npz")
    nodes, tris, rays = d[f"{tag}_nodes12"], d[f"{tag}_tris12"], d[f"{tag}_rays"]
    ren.upload_bvh(nodes, tris)
    eps, inf = float(d["eps"]), float(d["inf"])
    c = ren.debug_trace(0, rays[:, 0:3], rays[:, 4:7], eps=eps, inf=inf)
    a = ren.debug_trace(1, rays[:, 0:3], rays[:, 4:7], rays[:, 3], eps=eps, inf=inf)
    compared, hits, tie_rays = check_trace_kat(tag, d, lambda i: (c[i, 0] < np.float32(inf), c[i, 0], c[i, 1:4], c[i, 4:7]), lambda i: a[i, 0] != 0.0)
    assert compared >= 1000 and hits >= 900
    if tag == "crate":
        assert tie_rays >= 300


@pytest.mark.parametrize("tag", ["svgf", "plain", "motion", "svgf_moving"])
def test_hip_matches_reference_glsl_present(ren, tag):
    d = np.load(GOLDEN / "glsl_present_48x36.npz")
    pp = rt.RtPresentParams.from_buffer_copy(d[f"pp_{tag}"].tobytes())
    ren.resize(48, 36)
    for which, k in ((rt.RT_TARGET_COLOR, "color"), (rt.RT_TARGET_MOTION, "motion"), (rt.RT_TARGET_GPOS, "gpos"), (rt.RT_TARGET_GNRM, "gnrm")):
        ren.write_target(which, d[f"{k}_{tag}"])
    got = ren.present_with(pp)
    want = d[f"rgba_{tag}"]
    diff = np.abs(got.astype(np.int32) - want.astype(np.int32))
    assert diff.max() <= 1 and np.mean(got == want) >= 0.995, (tag, diff.max(), np.mean(got == want))


def test_write_target_round_trip(ren):
    rng = np.random.default_rng(1)
    ren.resize(70, 37)                                    # ragged: partial tiles on both axes
    for which, ch in ((rt.RT_TARGET_COLOR, 4), (rt.RT_TARGET_MOTION, 2), (rt.RT_TARGET_GPOS, 4), (rt.RT_TARGET_GNRM, 4)):
        img = rng.integers(0, 0x7C00, size=(37, 70, ch), dtype=np.uint16)
        ren.write_target(which, img)
        assert np.array_equal(ren.read_target(which), img)

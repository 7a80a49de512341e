"""Present pass (shaders/rt/rt_present.frag, the second half of renderRay): oracle KATs on the CPU and HIP-vs-oracle
parity on the GPU.  Output is RGBA8, so the bar is bit equality."""
import numpy as np
import pytest

import opengl_raytracing_amd as rt
import scenes


def _const_targets(orc, W, H, rgb, m2=0.0, motion=(0.0, 0.0)):
    def half(v):
        return np.array(v, np.float32).astype(np.float16).view(np.uint16)
    color = np.tile(half(list(rgb) + [m2]), (H, W, 1))
    mot = np.tile(half(list(motion)), (H, W, 1))
    gpos = np.tile(half([0.5, 0.25, -1.0, 1.0]), (H, W, 1))
    gnrm = np.tile(half([0.0, 1.0, 0.0, 0.0]), (H, W, 1))
    return [color, mot, gpos, gnrm]


def test_oracle_present_kats(orc):
    W, H = 12, 9
    p = orc.default_render_params()
    pp = rt.make_present_params(p, False, W, H)
    assert (pp.enableSVGF, pp.showMotion) == (1, 0) and abs(pp.varMax - 0.05) < 1e-7 and abs(pp.svgfStrength - 0.7) < 1e-7
    # a constant image is a fixed point of the filter: output = gamma(ACES(c)) everywhere (rt_present.frag:65-69, :263)
    c = np.array([0.25, 0.5, 1.0], np.float32)
    out = orc.present(pp, _const_targets(orc, W, H, c))
    x = c.astype(np.float16).astype(np.float64)
    aces = np.clip((x * (2.51 * x + 0.03)) / (x * (2.43 * x + 0.59) + 0.14), 0, 1) ** (1 / 2.2)
    want = np.rint(aces * 255).astype(int)
    assert np.abs(out[..., :3].astype(int) - want).max() <= 1 and (out[..., 3] == 255).all()
    assert (out == out[0, 0]).all()
    # SVGF off = raw colour through the same curve; black stays black; exposure scales before the curve
    pp2 = rt.make_present_params(p, False, W, H); pp2.enableSVGF = 0
    assert np.array_equal(orc.present(pp2, _const_targets(orc, W, H, c)), out)
    assert not orc.present(pp, _const_targets(orc, W, H, [0, 0, 0]))[..., :3].any()
    # motion view: zero motion is in the deadband -> black; +x motion -> hue 0.5 (cyan), value = clamp(|m| * scale)
    pm = rt.make_present_params(p, True, W, H)
    assert not orc.present(pm, _const_targets(orc, W, H, c))[..., :3].any()
    cy = orc.present(pm, _const_targets(orc, W, H, c, motion=(0.5, 0.0)))[0, 0]
    assert tuple(cy) == (0, 255, 255, 255)
    # uv = (fragCoord + 0.5) / size samples the texel to the upper right (clamped at the border): reference quirk, kept
    t = _const_targets(orc, W, H, [0, 0, 0])
    t[0][3, 4, :3] = np.array([1.0, 1.0, 1.0], np.float16).view(np.uint16)
    raw = orc.present(pp2, t)
    assert raw[2, 3, 0] > 200 and raw[3, 4, 0] == 0
    # deterministic transcendental helpers stay close to libm
    import math
    L = orc.lib()
    for v in np.linspace(-20, 3, 200):
        assert abs(L.orc_exp(float(v)) - math.exp(float(np.float32(v)))) <= 3e-6 * math.exp(float(np.float32(v))) + 1e-38   # exp2(x*log2e): error grows with |x|, as on GPUs
    for a in np.linspace(-3.1, 3.1, 100):
        assert abs(L.orc_atan2(math.sin(a), math.cos(a)) - a) < 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("use_bvh", [False, True])
def test_present_matches_oracle_on_gpu(orc, use_bvh):
    W, H = 200, 120
    nodes, tris = scenes.bunny_bvh(3)
    faces = scenes.tiny_env(8)
    p = rt.default_render_params()
    p.sppPerFrame = 2
    cam = scenes.camera("closeup" if use_bvh else "default", aspect=W / H)
    with rt.Renderer() as r:
        r.upload_bvh(nodes, tris)
        r.upload_env(faces)
        r.resize(W, H)
        prev, prev_vp = None, None
        for frame in range(3):
            if frame == 2:
                cam.pos[0] += 0.05          # moving frame: non-zero motion vectors steer kVar / kColor
            view, proj = orc.camera_view(cam), orc.camera_proj(cam)
            vp = orc.mat4_mul(proj, view)
            prev_vp = vp if prev_vp is None else prev_vp
            u = orc.make_uniforms(p, cam, view, vp, prev_vp, W, H, frame, orc.camera_moved(vp, prev_vp), use_bvh, False,
                                  nodes.shape[0], tris.shape[0], True)
            r.render_ray(p, cam, use_bvh=use_bvh)
            want_t, _ = orc.render(u, nodes, tris, faces, prev)
            for variant in ("svgf", "raw", "motion", "exposure"):
                q = p.copy()
                show = variant == "motion"
                if variant == "raw":
                    q.enableSVGF = 0
                if variant == "exposure":
                    q.exposure, q.svgfStrength, q.svgfKColor = 2.5, 1.0, 3.0
                got = r.present(q, show)
                want = orc.present(rt.make_present_params(q, show, W, H), want_t)
                assert np.array_equal(got, want), (use_bvh, frame, variant, int((got != want).sum()))
            prev, prev_vp = want_t[0], vp


@pytest.mark.gpu
def test_present_refused_on_tile_parallel_ranks():
    with rt.Renderer(rank=0, world_size=2) as r:
        r.resize(64, 48)
        with pytest.raises(rt.RtError) as e:
            r.present(rt.default_render_params())
        assert e.value.code == rt.RT_ERR_UNSUPPORTED

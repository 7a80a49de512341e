"""Randomised parity sweep: random meshes, cameras (inside / outside / grazing), toggles, spp, frame sizes (ragged tiles),
moving and static cameras -- every frame of both pipelines must equal the oracle bit for bit.  Seeds are fixed; set
RT_FUZZ_CASES to run more cases than the default dozen."""
import os

import numpy as np
import pytest

import opengl_raytracing_amd as rt
import scenes

pytestmark = pytest.mark.gpu
CASES = int(os.environ.get("RT_FUZZ_CASES", "12"))


def _case(seed):
    rng = np.random.default_rng(1000 + seed)
    subdiv = int(rng.integers(0, 4))
    v, f = rt.meshgen.bunny_standin(subdiv, seed=int(rng.integers(1 << 30)))
    v = v * rng.uniform(0.3, 2.0) + rng.normal(0, 0.3, 3).astype(np.float32)
    nodes, tris = rt.build_bvh(rt.gather_triangles(v, f))
    W, H = int(rng.integers(17, 90)), int(rng.integers(17, 70))
    p = rt.default_render_params()
    p.sppPerFrame = int(rng.choice([1, 1, 2, 3, 4, 5]))
    for name in ("enableGI", "enableAO", "enableTAA", "enableJitter", "sunEnabled", "skyEnabled", "pointLightEnabled", "enableEnvMap"):
        setattr(p, name, int(rng.random() < 0.75))
    p.aoSamples = int(rng.integers(1, 7))
    p.aoRadius = float(rng.uniform(0.05, 1.5))
    p.sunYaw, p.sunPitch = float(rng.uniform(-180, 180)), float(rng.uniform(-80, 80))
    p.pointLightPos[0], p.pointLightPos[1], p.pointLightPos[2] = [float(x) for x in rng.normal(0, 2.0, 3)]
    cams = []
    base = scenes.camera("closeup", aspect=W / H)
    centre = v.mean(0)
    kind = rng.integers(0, 4)
    for k in range(3):
        c = scenes.camera("closeup", aspect=W / H)
        if kind == 0:      # orbiting outside, looking roughly at the mesh
            c.pos[0], c.pos[1], c.pos[2] = [float(x) for x in centre + rng.normal(0, 1, 3) * 2.5]
        elif kind == 1:    # inside / very close
            c.pos[0], c.pos[1], c.pos[2] = [float(x) for x in centre + rng.normal(0, 0.2, 3)]
        else:              # default-ish
            c.pos[0], c.pos[1], c.pos[2] = base.pos[0] + float(rng.normal(0, 0.3)), base.pos[1] + float(rng.normal(0, 0.3)), base.pos[2] + float(rng.normal(0, 0.3))
        c.yaw, c.pitch = float(rng.uniform(-180, 180)) if kind < 2 else base.yaw + float(rng.normal(0, 10)), float(rng.uniform(-60, 60)) if kind < 2 else float(rng.normal(0, 10))
        c.fov = float(rng.uniform(30, 100))
        cams.append(c)
    moving = bool(rng.random() < 0.5)
    faces = scenes.tiny_env(int(rng.choice([1, 2, 5, 8])), seed=int(rng.integers(100))) if p.enableEnvMap else None
    return nodes, tris, faces, p, cams, moving, W, H


@pytest.mark.parametrize("seed", range(CASES))
def test_random_scene_matches_the_oracle_on_both_pipelines(orc, seed):
    nodes, tris, faces, p, cams, moving, W, H = _case(seed)
    for pipeline in (rt.RT_PIPELINE_WAVEFRONT, rt.RT_PIPELINE_MEGAKERNEL):
        with rt.Renderer(pipeline=pipeline) as r:
            r.upload_bvh(nodes, tris)
            r.upload_env(faces)
            r.resize(W, H)
            prev, prev_vp = None, None
            for frame in range(3):
                cam = cams[frame] if moving else cams[0]
                u = rt.frame_uniforms(p, cam, W, H, frame, True, nodes.shape[0], tris.shape[0], prev_vp=prev_vp, env_loaded=faces is not None)
                prev_vp = rt.mat4_mul(rt.camera_proj(cam), rt.camera_view(cam))
                r.render_frame(u)
                want, _ = orc.render(u, nodes, tris, faces, prev)
                got = r.read_all()
                for g, w_, name in zip(got, want, ("color", "motion", "gpos", "gnrm")):
                    assert np.array_equal(g, w_), (seed, pipeline, frame, name, int(np.sum(g != w_)))
                prev = want[0]


@pytest.mark.parametrize("seed", range(CASES))
def test_random_analytic_scene_matches_the_oracle(orc, seed):
    """The analytic scene (plane + spheres, glass / mirror / point-light sphere): random cameras, materials, lights, toggles."""
    rng = np.random.default_rng(5000 + seed)
    W, H = int(rng.integers(17, 80)), int(rng.integers(17, 60))
    p = rt.default_render_params()
    p.sppPerFrame = int(rng.choice([1, 2, 3]))
    for name in ("enableGI", "enableAO", "enableTAA", "enableJitter", "sunEnabled", "skyEnabled", "pointLightEnabled", "enableEnvMap",
                 "matGlassEnabled", "matMirrorEnabled"):
        setattr(p, name, int(rng.random() < 0.75))
    p.matGlassIOR = float(rng.uniform(1.0, 2.2))
    p.matGlassDistortion = float(rng.uniform(0.0, 0.5))
    p.matMirrorGloss = float(rng.uniform(0.0, 1.0))
    p.matAlbedoGloss = float(rng.uniform(1.0, 128.0))
    p.matAlbedoSpecStrength = float(rng.uniform(0.0, 1.0))
    p.pointLightPos[0], p.pointLightPos[1], p.pointLightPos[2] = float(rng.normal(0, 2)), float(rng.uniform(0.2, 4)), float(rng.normal(0, 2))
    p.aoSamples = int(rng.integers(1, 6))
    faces = scenes.tiny_env(int(rng.choice([1, 3, 8])), seed=int(rng.integers(100))) if p.enableEnvMap else None
    cams = []
    for k in range(3):
        c = scenes.camera("default", aspect=W / H)
        c.pos[0] += float(rng.normal(0, 1.5)); c.pos[1] = float(rng.uniform(0.1, 5.0)); c.pos[2] += float(rng.normal(0, 1.5))
        c.yaw += float(rng.normal(0, 40)); c.pitch += float(rng.normal(0, 25)); c.fov = float(rng.uniform(25, 110))
        cams.append(c)
    moving = bool(rng.random() < 0.5)
    with rt.Renderer() as r:
        r.upload_env(faces)
        r.resize(W, H)
        prev, prev_vp = None, None
        for frame in range(3):
            cam = cams[frame] if moving else cams[0]
            u = rt.frame_uniforms(p, cam, W, H, frame, False, prev_vp=prev_vp, env_loaded=faces is not None)
            prev_vp = rt.mat4_mul(rt.camera_proj(cam), rt.camera_view(cam))
            r.render_frame(u)
            want, _ = orc.render(u, None, None, faces, prev)
            for g, w_, name in zip(r.read_all(), want, ("color", "motion", "gpos", "gnrm")):
                assert np.array_equal(g, w_), (seed, frame, name, int(np.sum(g != w_)))
            prev = want[0]


@pytest.mark.parametrize("seed", range(max(CASES // 3, 4)))
def test_random_tile_parallel_frames_equal_the_single_rank_frames(seed):
    """Random world sizes (2..8 ranks emulated on one GPU), ragged frame sizes, static or moving camera (with the history exchange):
    the assembled COLOR0 and the gathered present must equal the single-context results bit for bit."""
    import torch
    from opengl_raytracing_amd.dist_gather import wrap_device_bytes
    from test_gpu_multirank import _gather_blocks
    nodes, tris, faces, p, cams, moving, W, H = _case(7000 + seed)
    rng = np.random.default_rng(9000 + seed)
    world = int(rng.integers(2, 9))
    W, H = W + 40, H + 30
    for c in cams:
        c.aspect = W / H
    with rt.Renderer() as single:
        single.upload_bvh(nodes, tris); single.upload_env(faces); single.resize(W, H)
        ranks = [rt.Renderer(rank=r, world_size=world) for r in range(world)]
        try:
            for r in ranks:
                r.upload_bvh(nodes, tris); r.upload_env(faces); r.resize(W, H)
            prev_vp = None
            for frame in range(3):
                cam = cams[frame] if moving else cams[0]
                u = rt.frame_uniforms(p, cam, W, H, frame, True, nodes.shape[0], tris.shape[0], prev_vp=prev_vp, env_loaded=faces is not None)
                prev_vp = rt.mat4_mul(rt.camera_proj(cam), rt.camera_view(cam))
                single.render_frame(u)
                for r in ranks:
                    r.render_frame(u)
                allc = _gather_blocks(ranks, rt.RT_TARGET_COLOR)
                if moving:
                    for r in ranks:
                        ptr, n = r.history_exchange_buffer()
                        wrap_device_bytes(ptr, n, torch.device("cuda", 0)).copy_(allc.reshape(-1))
                        torch.cuda.synchronize()
                        r.history_exchanged()
                out = torch.empty((H, W, 8), dtype=torch.uint8, device="cuda")
                ranks[0].assemble_gathered(rt.RT_TARGET_COLOR, allc.data_ptr(), out.data_ptr())
                ranks[0].synchronize()
                want = single.read_target(rt.RT_TARGET_COLOR)
                assert np.array_equal(out.cpu().numpy().view("<u2").reshape(want.shape), want), (seed, world, frame, moving)
            g = [_gather_blocks(ranks, which) for which in range(4)]
            pp = rt.make_present_params(p, False, W, H)
            assert np.array_equal(ranks[0].present_gathered(pp, *[t.data_ptr() for t in g]), single.present_with(pp)), (seed, world)
        finally:
            for r in ranks:
                r.close()


@pytest.mark.parametrize("W,H", [(1, 1), (16, 16), (15, 17), (33, 1), (1, 40), (257, 3)])
def test_degenerate_frame_sizes(orc, W, H):
    """One pixel, exactly one tile, ragged in both directions, single rows / columns -- BVH on both pipelines and the analytic scene."""
    nodes, tris = scenes.bunny_bvh(2)
    faces = scenes.tiny_env(4)
    p = rt.default_render_params()
    p.sppPerFrame = 2
    cam = scenes.camera("closeup", aspect=W / H)
    for use_bvh, pipeline in ((True, rt.RT_PIPELINE_WAVEFRONT), (True, rt.RT_PIPELINE_MEGAKERNEL), (False, rt.RT_PIPELINE_AUTO)):
        with rt.Renderer(pipeline=pipeline) as r:
            r.upload_bvh(nodes, tris)
            r.upload_env(faces)
            r.resize(W, H)
            prev = None
            for frame in range(2):
                u = rt.frame_uniforms(p, cam, W, H, frame, use_bvh, nodes.shape[0], tris.shape[0])
                r.render_frame(u)
                want, _ = orc.render(u, nodes, tris, faces, prev)
                for g, w_, name in zip(r.read_all(), want, ("color", "motion", "gpos", "gnrm")):
                    assert np.array_equal(g, w_), (W, H, use_bvh, pipeline, frame, name)
                prev = want[0]
            pp = rt.make_present_params(p, False, W, H)
            assert np.array_equal(r.present_with(pp), orc.present(pp, r.read_all()))


def test_empty_and_single_triangle_bvh(orc):
    """uUseBVH = 1 with no triangles (rt_bvh.glsl:194: traceBVH returns false) and with a one-leaf tree."""
    W, H = 40, 24
    p = rt.default_render_params()
    cam = scenes.camera("closeup", aspect=W / H)
    tri = np.array([[-2.6, 1.0, -0.5, -1.4, 1.1, 0.4, -2.0, 2.2, 0.0]], np.float32)
    nodes1, tris1 = rt.build_bvh(tri)
    for nodes, tris in ((np.zeros((0, 12), np.float32), np.zeros((0, 12), np.float32)), (nodes1, tris1)):
        for pipeline in (rt.RT_PIPELINE_AUTO, rt.RT_PIPELINE_MEGAKERNEL):
            with rt.Renderer(pipeline=pipeline) as r:
                r.upload_bvh(nodes, tris)
                r.resize(W, H)
                prev = None
                for frame in range(2):
                    u = rt.frame_uniforms(p, cam, W, H, frame, True, nodes.shape[0], tris.shape[0], env_loaded=False)
                    r.render_frame(u)
                    want, cnt = orc.render(u, nodes if nodes.shape[0] else None, tris if tris.shape[0] else None, None, prev)
                    for g, w_, name in zip(r.read_all(), want, ("color", "motion", "gpos", "gnrm")):
                        assert np.array_equal(g, w_), (nodes.shape[0], pipeline, frame, name)
                    prev = want[0]
    assert cnt.hitPixels > 0          # the single triangle is in view


@pytest.mark.parametrize("seed", range(CASES))
def test_random_scene_batched_frames_equal_frame_by_frame_and_the_oracle(orc, seed):
    """rt_render_frames on the random scenes (static camera): a random split of 9 frames into batches == the same frames one by one,
    on one rank and on a random rank of a random world; the last frame of the single-rank run against the oracle too."""
    nodes, tris, faces, p, cams, _, W, H = _case(seed)
    rng = np.random.default_rng(9000 + seed)
    cam = cams[0]
    us = [rt.frame_uniforms(p, cam, W, H, f, True, nodes.shape[0], tris.shape[0], env_loaded=faces is not None) for f in range(9)]
    cuts = sorted(set(int(x) for x in rng.integers(1, 9, size=int(rng.integers(0, 4)))))
    parts = [us[a:b] for a, b in zip([0] + cuts, cuts + [9])]
    world = int(rng.integers(1, 5))
    for wsize, rank in ((1, 0), (world, int(rng.integers(0, world)))):
        with rt.Renderer(rank=rank, world_size=wsize) as a, rt.Renderer(rank=rank, world_size=wsize) as b:
            for r in (a, b):
                r.upload_bvh(nodes, tris)
                r.upload_env(faces)
                r.resize(W, H)
            for u in us:
                a.render_frame(u)
            for part in parts:
                b.render_frames(part)
            assert a.frame_index == b.frame_index == 9
            ga, gb = a.read_all(), b.read_all()
            for x, y in zip(ga, gb):
                assert np.array_equal(x, y), (seed, wsize, rank, [len(q) for q in parts])
            if wsize == 1:
                prev = None
                for u in us:
                    want, _ = orc.render(u, nodes, tris, faces, prev)
                    prev = want[0]
                for x, y in zip(gb, want):
                    assert np.array_equal(x, y), (seed, "oracle")

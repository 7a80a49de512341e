"""Tile-parallel path on ONE GPU (the box has a single MI355X): two contexts act as ranks 0 and 1 of a
world of 2 and must reproduce the single-context frame after the gather block exchange + un-tiling
kernel; the torch plumbing bench.py uses (zero-copy pointer wrapping, ExternalStream, a world-size-1
RCCL group) is exercised too.  The real multi-process RCCL transport is covered on CPU by
tests/test_tiles_gloo.py (gloo, world_size 2)."""
import os

import numpy as np
import pytest

import opengl_raytracing_amd as rt
import scenes

pytestmark = pytest.mark.gpu


def _scene(W, H, spp=2):
    nodes, tris = scenes.bunny_bvh(4)
    faces = scenes.tiny_env(16)
    p = rt.default_render_params()
    p.sppPerFrame = spp
    cam = scenes.camera("closeup", aspect=W / H)
    return nodes, tris, faces, p, cam


def _setup(r, nodes, tris, faces, W, H):
    r.upload_bvh(nodes, tris)
    r.upload_env(faces)
    r.resize(W, H)


@pytest.mark.parametrize("world", [2, 3])
def test_ranks_on_one_gpu_assemble_to_the_single_rank_frame(world):
    import torch
    from opengl_raytracing_amd.dist_gather import wrap_device_bytes
    W, H = 200, 120
    nodes, tris, faces, p, cam = _scene(W, H)
    with rt.Renderer() as single:
        _setup(single, nodes, tris, faces, W, H)
        ranks = [rt.Renderer(rank=r, world_size=world) for r in range(world)]
        try:
            for r in ranks:
                _setup(r, nodes, tris, faces, W, H)
            for frame in range(3):
                u = rt.frame_uniforms(p, cam, W, H, frame, True, nodes.shape[0], tris.shape[0])
                single.render_frame(u)
                for r in ranks:
                    r.render_frame(u)
                for which in range(4):
                    want = single.read_target(which)
                    blk = ranks[0].gather_block_bytes(which)
                    gathered = torch.empty((world, blk), dtype=torch.uint8, device="cuda")
                    for i, r in enumerate(ranks):
                        r.synchronize()
                        ptr, n = r.local_target(which)
                        assert n == blk
                        gathered[i].copy_(wrap_device_bytes(ptr, n, torch.device("cuda", 0)))
                    out = torch.empty((H, W, rt.TARGET_CHANNELS[which] * 2), dtype=torch.uint8, device="cuda")
                    torch.cuda.synchronize()
                    ranks[0].assemble_gathered(which, gathered.data_ptr(), out.data_ptr())
                    ranks[0].synchronize()
                    got = out.cpu().numpy().view("<u2").reshape(want.shape)
                    assert np.array_equal(got, want), (frame, which)
                    # a rank's own readback: its pixels match, everything else is zero
                    from opengl_raytracing_amd import tiles
                    for i, r in enumerate(ranks):
                        mine = tiles.owner_mask(W, H, i, world).astype(bool)
                        loc = r.read_target(which)
                        assert np.array_equal(loc[mine], want[mine]) and not loc[~mine].any()
        finally:
            for r in ranks:
                r.close()


def test_moving_camera_is_refused_with_more_than_one_rank():
    W, H = 64, 48
    nodes, tris, faces, p, cam = _scene(W, H)
    with rt.Renderer(rank=0, world_size=2) as r:
        _setup(r, nodes, tris, faces, W, H)
        u = rt.frame_uniforms(p, cam, W, H, 0, True, nodes.shape[0], tris.shape[0])
        u.cameraMoved = 1
        with pytest.raises(rt.RtError) as e:
            r.render_frame(u)
        assert e.value.code == rt.RT_ERR_UNSUPPORTED


def test_frame_gatherer_plumbing_on_a_one_rank_rccl_group():
    import torch
    import torch.distributed as dist
    from opengl_raytracing_amd.dist_gather import FrameGatherer
    W, H = 160, 96
    nodes, tris, faces, p, cam = _scene(W, H)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        with rt.Renderer() as r:
            _setup(r, nodes, tris, faces, W, H)
            g = FrameGatherer(r)
            for frame in range(3):
                r.render_frame(rt.frame_uniforms(p, cam, W, H, frame, True, nodes.shape[0], tris.shape[0]))
                g.gather()
                assert np.array_equal(g.frame_halfs(), r.read_target(rt.RT_TARGET_COLOR))
        # all_reduce of the timing scalar as bench.py does
        t = torch.tensor([1.5], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        assert t.item() == 1.5
    finally:
        dist.destroy_process_group()

"""Multi-GPU path on the CPU: world_size-2 `gloo` process group.  Each rank renders only its tiles
(with the oracle standing in for the device -- this is a test of the partition / gather / assemble
logic, not of the kernels), packs them into the tile-major block the library exposes through
rt_local_target, rank 0 gathers and assembles, and the result must equal the single-rank frame."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent
GOLDEN = ROOT / "tests" / "golden"


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_path):
    for p in (str(ROOT), str(ROOT / "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import opengl_raytracing_amd as rt
    from opengl_raytracing_amd import tiles
    import oracle as orc
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    d = np.load(GOLDEN / "bvh_closeup_48x32.npz")
    H, W = d["color0"].shape[:2]
    prev_full = None
    frames = []
    for f in range(d["uniforms"].shape[0]):
        u = rt.RtUniforms.from_buffer_copy(d["uniforms"][f].tobytes())
        mask = tiles.owner_mask(W, H, rank, world)
        # a rank only ever reads its own history (static camera): feed it just its own pixels of the previous frame
        prev = None if prev_full is None else prev_full * mask[..., None].astype(np.uint16)
        outs, _ = orc.render(u, d["nodes12"], d["tris12"], d["env"], prev, mask=mask, nthreads=2)
        local = torch.from_numpy(tiles.pack_local(outs[0], rank, world).view(np.uint8).copy())   # bytes: every backend moves uint8
        gathered = [torch.empty_like(local) for _ in range(world)] if rank == 0 else None
        dist.gather(local, gathered, dst=0)
        full = None
        if rank == 0:
            full = tiles.assemble([g.numpy().view(np.uint16).reshape(-1, 4) for g in gathered], W, H)
            frames.append(full)
        # every rank needs its own previous pixels next frame; broadcast the assembled frame (stand-in for keeping local history)
        t = torch.from_numpy(full.view(np.uint8).copy()) if rank == 0 else torch.empty((H, W, 8), dtype=torch.uint8)
        dist.broadcast(t, src=0)
        prev_full = t.numpy().view(np.uint16).reshape(H, W, 4).copy()
    if rank == 0:
        np.save(out_path, np.stack(frames))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_two_rank_tile_gather_equals_single_rank(tmp_path, world):
    """world 2 and 3: the 48x32 frame is six tiles -- three / two per rank, dealt round-robin."""
    out = tmp_path / "frames.npy"
    mp.spawn(_worker, args=(world, _free_port(), str(out)), nprocs=world, join=True)
    frames = np.load(out)
    d = np.load(GOLDEN / "bvh_closeup_48x32.npz")
    for f in range(frames.shape[0]):
        assert np.array_equal(frames[f], d[f"color{f}"]), f


def _worker_moving(rank, world, port, out_path):
    """Moving camera: reprojection reads the previous frame in other ranks' tiles, so every rank ALL-GATHERS the COLOR0 blocks
    after each frame (what FrameGatherer(exchange_history=True) does with RCCL into rt_history_exchange_buffer)."""
    for p in (str(ROOT), str(ROOT / "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import opengl_raytracing_amd as rt
    from opengl_raytracing_amd import tiles
    import oracle as orc
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    d = np.load(GOLDEN / "glsl_analytic_moving_48x36.npz")
    H, W = d["color0"].shape[:2]
    prev_full, frames = None, []
    for f in range(d["uniforms"].shape[0]):
        u = rt.RtUniforms.from_buffer_copy(d["uniforms"][f].tobytes())
        mask = tiles.owner_mask(W, H, rank, world)
        outs, _ = orc.render(u, None, None, d["env"], prev_full, mask=mask, nthreads=2)
        local = torch.from_numpy(tiles.pack_local(outs[0], rank, world).view(np.uint8).copy())
        blocks = [torch.empty_like(local) for _ in range(world)]
        dist.all_gather(blocks, local)                                         # every rank gets every block
        prev_full = tiles.assemble([b.numpy().view(np.uint16).reshape(-1, 4) for b in blocks], W, H)
        frames.append(prev_full)
    if rank == 1:                                                              # any rank holds the whole frame now
        np.save(out_path, np.stack(frames))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_all_gather_for_a_moving_camera(tmp_path, orc):
    import opengl_raytracing_amd as rt
    out = tmp_path / "frames_moving.npy"
    mp.spawn(_worker_moving, args=(2, _free_port(), str(out)), nprocs=2, join=True)
    got = np.load(out)
    d = np.load(GOLDEN / "glsl_analytic_moving_48x36.npz")
    prev = None
    for f in range(d["uniforms"].shape[0]):
        u = rt.RtUniforms.from_buffer_copy(d["uniforms"][f].tobytes())
        assert f == 0 or u.cameraMoved == 1
        want, _ = orc.render(u, None, None, d["env"], prev)
        assert np.array_equal(got[f], want[0]), f
        prev = want[0]

"""Tile-parallel frame assembly across GPUs: one process per GPU.

Two ways to run the exchange:
* NativeGatherer -- the library's own RCCL communicator behind the C ABI (rt_comm_init / rt_gather_frame /
  rt_exchange_history); torch.distributed is only used to hand rank 0's 128-byte id to the other ranks.  This is what
  bench.py and a C++ host (rt_cli --ranks N) use.
* FrameGatherer  -- the same exchange issued from Python through torch.distributed (RCCL) on the library's device pointers
  and streams; kept as the independent cross-check of the native path and as bench.py's fall-back.

Each rank renders its 16x16 tiles into tile-major local targets (csrc/rt_frame.hpp).  Per frame the
only exchange is ONE gather of the chosen target to rank 0 (RCCL lowers it to grouped send/recv
over the point-to-point xGMI links, so the root's 7 inbound links run in parallel), followed by
the un-tiling kernel on rank 0.  Nothing is reduced, so no all-reduce / ring is involved.
torch is plumbing here: it wraps the library's device pointers (no copies) and orders the
collective on the library's HIP stream.
"""
from __future__ import annotations

import torch
import torch.distributed as dist

from . import RT_TARGET_COLOR, TARGET_CHANNELS


class _DevBytes:
    """Expose a raw device pointer to torch.as_tensor (CUDA array interface, zero copy)."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


def wrap_device_bytes(ptr, nbytes, device):
    return torch.as_tensor(_DevBytes(ptr, nbytes), device=device)


class FrameGatherer:
    """exchange_history: also all-gather COLOR0 to EVERY rank after each frame (rt_history_exchange_buffer), which a moving
    camera needs: reprojection reads the previous frame in other ranks' tiles.  Costs worldSize-1 more block copies per rank and
    frame over xGMI; leave it off for static cameras (a pixel then only reads its own history)."""

    def __init__(self, renderer, which=RT_TARGET_COLOR, group=None, exchange_history=False, gather_every=1):
        self.ren, self.which, self.group = renderer, which, group
        self.exchange_history = exchange_history
        self.gather_every = max(1, int(gather_every))
        self.frames = 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.device = torch.device("cuda", torch.cuda.current_device())
        # a gloo group (rehearsals with several ranks on ONE GPU, which RCCL refuses): blocks are staged through host memory
        self.host_staged = bool(dist.is_initialized() and dist.get_backend(group) == "gloo")
        self._streams = {}
        self.block = renderer.gather_block_bytes(which)
        ch = TARGET_CHANNELS[which]
        # one (gathered, frame) buffer pair per frame lane = per stream: gathers of consecutive frames run on different,
        # unordered streams, so a shared pair would let frame f+1's gather overwrite what frame f's un-tiling still reads
        self._bufs = {}
        self._ch = ch
        self.gathered = self.frame = None       # the pair of the most recent gather()
        self._wrapped = {}
        self.timed = False                      # bench.py: device time of every gather (events on the gather's stream)
        self._events = []
        self._bytes = 0

    def _pair(self, sp):
        b = self._bufs.get(sp)
        if b is None and self.rank == 0:
            b = self._bufs[sp] = (torch.empty((self.world, self.block), dtype=torch.uint8, device=self.device),   # bytes: RCCL has no int16
                                  torch.empty((self.ren.height, self.ren.width, self._ch * 2), dtype=torch.uint8, device=self.device))
        return b if b is not None else (None, None)

    def _local(self):
        ptr, nbytes = self.ren.local_target(self.which)     # COLOR0 ping-pongs between two buffers
        t = self._wrapped.get(ptr)
        if t is None:
            t = self._wrapped[ptr] = wrap_device_bytes(ptr, nbytes, self.device)
        return t

    def gather(self):
        """Enqueue gather + assemble behind the frame just rendered (asynchronous; same stream as the kernels)."""
        loc = self._local()
        sp = self.ren.stream()                 # consecutive frames alternate between two streams: follow the last frame's
        stream = self._streams.get(sp)
        if stream is None:
            stream = self._streams[sp] = torch.cuda.ExternalStream(sp, device=self.device)
        self.gathered, self.frame = self._pair(sp)
        with torch.cuda.stream(stream):
            if self.timed:
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                ev[0].record(stream)
                self._events.append(ev)
                self._bytes += self.block * (self.world - 1) if self.rank == 0 else self.block
            if self.world > 1 and self.host_staged:
                h = loc.cpu()                    # waits for the frame on this stream
                parts = [torch.empty_like(h) for _ in range(self.world)] if self.rank == 0 else None
                dist.gather(h, parts, dst=0, group=self.group)
                if self.rank == 0:
                    self.gathered.copy_(torch.stack(parts))
            elif self.world > 1:
                dist.gather(loc, list(self.gathered.unbind(0)) if self.rank == 0 else None, dst=0, group=self.group)
            elif self.rank == 0:
                self.gathered[0].copy_(loc, non_blocking=True)
            if self.rank == 0:
                self.ren.assemble_gathered(self.which, self.gathered.data_ptr(), self.frame.data_ptr())
            if self.exchange_history and self.world > 1:
                col = loc if self.which == RT_TARGET_COLOR else self._local_of(RT_TARGET_COLOR)
                ptr, nbytes = self.ren.history_exchange_buffer()
                dst = self._wrapped.get(("hist", ptr))
                if dst is None:
                    dst = self._wrapped[("hist", ptr)] = wrap_device_bytes(ptr, nbytes, self.device)
                dist.all_gather_into_tensor(dst, col, group=self.group)
            if self.timed:
                self._events[-1][1].record(stream)
        if self.exchange_history and self.world > 1:
            self.ren.history_exchanged()       # the next frame's temporal resolve now waits for the all-gather too
        return self.frame

    def timing(self):
        """Device time of the gathers since `timed` was set (synchronises): {"ms", "gathers", "bytes"}; bytes = received on rank 0, sent elsewhere."""
        torch.cuda.synchronize()
        ms = sum(a.elapsed_time(b) for a, b in self._events)
        out = {"ms": ms, "gathers": len(self._events), "bytes": self._bytes}
        self._events, self._bytes = [], 0
        return out

    def after(self, n_frames=1, last=False):
        """n_frames were rendered since the previous call (one rt_render_frames batch): gather the newest one if a gather_every boundary
        was crossed, or if `last`.  -> whether a gather was enqueued."""
        before = self.frames
        self.frames += n_frames
        if last or before // self.gather_every != self.frames // self.gather_every:
            self.gather()
            return True
        return False

    def _local_of(self, which):
        ptr, nbytes = self.ren.local_target(which)
        t = self._wrapped.get(ptr)
        if t is None:
            t = self._wrapped[ptr] = wrap_device_bytes(ptr, nbytes, self.device)
        return t

    def gather_targets(self):
        """Rank 0: all four targets of the last frame as gathered-block arrays (for rt_present_gathered); others: None."""
        sp = self.ren.stream()
        stream = self._streams.get(sp) or self._streams.setdefault(sp, torch.cuda.ExternalStream(sp, device=self.device))
        outs = []
        with torch.cuda.stream(stream):
            for which in range(4):
                loc = self._local_of(which)
                blk = self.ren.gather_block_bytes(which)
                g = torch.empty((self.world, blk), dtype=torch.uint8, device=self.device) if self.rank == 0 else None
                if self.world > 1:
                    dist.gather(loc, list(g.unbind(0)) if self.rank == 0 else None, dst=0, group=self.group)
                else:
                    g[0].copy_(loc, non_blocking=True)
                outs.append(g)
        return outs if self.rank == 0 else None

    def present(self, present_params):
        """Present pass of the last frame on rank 0 (RGBA8 [H, W, 4]); other ranks take part in the gathers and return None."""
        g = self.gather_targets()
        if self.rank != 0:
            return None
        self.ren.synchronize()
        torch.cuda.synchronize()
        return self.ren.present_gathered(present_params, *[t.data_ptr() for t in g])

    def frame_halfs(self):
        """Rank 0: the assembled frame as uint16 half bit patterns [H, W, C] on the host (synchronises)."""
        self.ren.synchronize()
        torch.cuda.synchronize()
        ch = TARGET_CHANNELS[self.which]
        return self.frame.cpu().numpy().view("<u2").reshape(self.ren.height, self.ren.width, ch)


class NativeGatherer:
    """The exchange through the library's own communicator.  gather_every=k: only every k-th frame is gathered (static camera:
    the accumulation history is tile-local, rt_taa.glsl:86-105, so the frames in between need no exchange)."""

    def __init__(self, renderer, which=RT_TARGET_COLOR, group=None, exchange_history=False, gather_every=1):
        from . import comm_unique_id
        self.ren, self.which, self.exchange_history = renderer, which, exchange_history
        self.gather_every = max(1, int(gather_every))
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        # every rank first proves that it can load RCCL and make an id (only rank 0's is used); the outcome is agreed on BEFORE
        # the collective ncclCommInitRank, so that one rank without a usable librccl cannot leave the others waiting in it
        err = None
        try:
            my_id = comm_unique_id()
        except Exception as e:   # noqa: BLE001
            my_id, err = None, e
        if self.world > 1 or dist.is_initialized():
            ok = torch.tensor([0.0 if err else 1.0], device=torch.device("cuda", torch.cuda.current_device()))
            dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
            if ok.item() == 0:
                raise RuntimeError(f"RCCL is not usable on every rank (this rank: {err!r})")
        elif err:
            raise err
        ids = [my_id if self.rank == 0 else None]
        if self.world > 1:
            dist.broadcast_object_list(ids, src=0, group=group)      # 128 bytes over the control plane
        renderer.comm_init(ids[0])
        self.frames = 0

    def gather(self, force=False):
        """After render_frame: gather this frame if it is due (or force=True).  Asynchronous."""
        return self.after(1, last=force)

    def after(self, n_frames=1, last=False):
        """n_frames were rendered since the previous call (one rt_render_frames batch): gather the newest one if a gather_every boundary
        was crossed, or if `last`.  -> whether a gather was enqueued."""
        before = self.frames
        self.frames += n_frames
        if self.exchange_history and self.world > 1:
            self.ren.exchange_history()
        if last or before // self.gather_every != self.frames // self.gather_every:
            self.ren.gather_frame(self.which)
            return True
        return False

    def frame_halfs(self):
        return self.ren.read_gathered(self.which)

    def present(self, present_params):
        for which in range(4):
            self.ren.gather_frame(which)
        if self.rank != 0:
            self.ren.synchronize()
            return None
        return self.ren.present_last_gathered(present_params)

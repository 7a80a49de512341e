#!/usr/bin/env python3
"""bench.py -- Mray/s of the ray-trace hot path on MI355X (BASELINE.json metric).

One "step" = one frame of the hot path (rt_render_frame through the C ABI) over the synthetic
workload of BASELINE.json configs[1]: procedural bunny stand-in (81 920 triangles, median-split
BVH), 1920x1080, 4 spp, one bounce (GI) + AO, Sky_01 environment, default RenderParams, static
camera, frame indices continuing from the warm-up.  Inputs are resident in HBM before the timed
region.  Headline camera = the close-up of SURVEY.md 8d (mesh ~45 % of the frame); the
reference's default camera (mesh < 1 % of the frame) is reported beside it in "default_camera".

N > 1 (launched by torch.distributed.run, one process per GPU): the frame's 16x16 tiles are dealt
round-robin to the ranks (the frame is fixed, so this is STRONG scaling), every rank holds a BVH
replica, and each frame ends with one RCCL gather of COLOR0 to rank 0 over xGMI plus
the un-tiling kernel.  value = rays of the whole frame / max-over-ranks time.

A "ray" is one traceBVH / traceBVHShadow call of the reference's shader for this frame
(SURVEY.md 8d), counted by the library's work counters in a separate, untimed pass and checked
against the oracle's count on the CPU sample.
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
for p in (str(ROOT), str(ROOT / "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

# Ceiling of the vector L1's gather path in G lane-loads/s (16 B per lane and load), per access shape, from tools/gather.hip with the
# table L2-resident and 20 waves/CU (profiles/r02_gather_microbench_pmc.txt: the PMC passes show TCP_TOTAL_CACHE_ACCESSES = one per
# divergent lane-load, processed at 0.8-1.0 per clock and CU): 64 lanes x 8 x dwordx4 from one 128-byte record per lane (the any-hit
# kernel's 4-wide node) 1.05e9 loads in 1.502 ms; 64 lanes x 4 x dwordx4 from a 64-byte record (2-wide node) 5.24e8 in 0.888 ms.
L1_GATHER_PEAK_G = {"trace_shadow": 1.05e9 / 1.502e-3 / 1e9, "trace_gi": 5.24e8 / 0.888e-3 / 1e9, "trace_primary": 5.24e8 / 0.888e-3 / 1e9}
HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
W, H, SPP = 1920, 1080, 4


def algorithmic_bytes(c, npix):
    """SURVEY.md 8d / BASELINE.md: reference-layout bytes of one frame."""
    return 48 * c.nodeFetch + 48 * c.triFetch + npix * 36 + 12 * c.envLookup


def kernel_source_sha():
    """sha256 over the device sources: ties a committed PMC traffic figure to the kernels it was measured on."""
    import hashlib
    h = hashlib.sha256()
    for f in sorted((ROOT / "opengl-raytracing_amd" / "csrc").glob("*.h*")):
        h.update(f.name.encode())
        h.update(f.read_bytes())
    return h.hexdigest()


def usable_cores():
    """CPU threads this process may really use: affinity, capped by the cgroup CPU quota (the GPU box gives 16 of 256)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200, help="timed frames (default 200: a timed region of ~0.4 s)")
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--subdiv", type=int, default=6, help="icosphere subdivisions of the bunny stand-in (6 = 81 920 tris)")
    ap.add_argument("--pipeline", default="auto", choices=["auto", "mega", "wave"])
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="budget of EACH oracle (CPU baseline) sample -- one thread, then all cores; 0 = skip")
    ap.add_argument("--no-default-camera", action="store_true")
    ap.add_argument("--batch", type=int, default=8, help="frames per rt_render_frames call: consecutive frames of the static camera share one set of "
                    "kernel launches (bit-identical to frame-by-frame rendering, tests/test_gpu_baseline_configs.py); 1 = one rt_render_frame per step")
    ap.add_argument("--no-frame-by-frame", action="store_true", help="skip the unbatched comparison pass (config.ms_per_step_frame_by_frame); used under "
                    "rocprofv3 so that the kernel statistics hold batched launches only")
    ap.add_argument("--gather-every", type=int, default=1, help="N > 1 GPUs: gather COLOR0 to rank 0 every k-th frame (1 = every frame; a static "
                    "camera's history is tile-local, so BASELINE configs[4] needs one gather per 32 accumulated frames)")
    ap.add_argument("--force-gather", action="store_true", help="rehearsal on one GPU: run the N > 1 code path (process group, communicator, "
                    "gather per frame) with a world of one; launch with torch.distributed.run --nproc-per-node 1")
    ap.add_argument("--gather", default="native", choices=["native", "torch"], help="N > 1: the library's own RCCL communicator (C ABI) or torch.distributed")
    # other BASELINE.json configurations, for side measurements (the default line is configs[1], the one `metric` is quoted on)
    ap.add_argument("--size", default="1920x1080", help="framebuffer WxH (configs[3]: 3840x2160)")
    ap.add_argument("--spp", type=int, default=4, help="samples per pixel and frame (configs[2-3]: 16, configs[4]: 64)")
    ap.add_argument("--scene", default="bunny", choices=["bunny", "1m"], help="1m = configs[4]'s 1M-triangle multi-object scene")
    ap.add_argument("--hybrid", action="store_true", help="EXTENSION (not in the reference; SURVEY 8d config 3 run B): the analytic scene "
                    "(floor, glass / mirror / diffuse spheres) with the mesh added to it, reference default camera, megakernel")
    ap.add_argument("--gi-bounces", type=int, default=1, help="EXTENSION: diffuse bounces of the analytic / hybrid GI path (configs[2]: 4)")
    args = ap.parse_args()
    global W, H, SPP
    W, H = (int(v) for v in args.size.lower().split("x"))
    SPP = args.spp

    import torch
    import torch.distributed as dist
    import opengl_raytracing_amd as rt
    from opengl_raytracing_amd.dist_gather import FrameGatherer, NativeGatherer
    import scenes

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC for RCCL between processes: before anything initialises HIP
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus N>1 must be launched with torch.distributed.run (one process per GPU)")
        args.gpus = world
    torch.cuda.set_device(local_rank)
    multi = world > 1 or args.force_gather      # the tile-parallel code path (process group, communicator, gathers)
    if multi:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    pipeline = {"auto": rt.RT_PIPELINE_AUTO, "mega": rt.RT_PIPELINE_MEGAKERNEL, "wave": rt.RT_PIPELINE_WAVEFRONT}[args.pipeline]
    if args.scene == "1m":
        v, fidx = rt.meshgen.million_triangle_scene()
        nodes, tris = rt.build_bvh(rt.gather_triangles(v, fidx, np.eye(4, dtype=np.float32).reshape(-1)))
    else:
        nodes, tris = scenes.bunny_bvh(args.subdiv)
    faces = scenes.env_faces("Sky_01")
    params = rt.default_render_params()
    params.sppPerFrame = SPP
    npix = W * H

    gather_path = {"path": "library-owned RCCL communicator (rt_comm_init / rt_gather_frame)", "gather_every": args.gather_every,
                   "when": "after each batch of frames (its last frame), and after the last timed frame"}

    def make_renderer(count):
        # work counters (reference units) come from the reference-shaped megakernel; the timed run uses `pipeline`
        r = rt.Renderer(device=local_rank, rank=rank, world_size=world,
                        pipeline=rt.RT_PIPELINE_MEGAKERNEL if count else pipeline, count_work=count)
        r.upload_bvh(nodes, tris)
        r.upload_env(faces)
        r.resize(W, H)
        if args.gi_bounces != 1:
            r.set_extension(gi_bounces=args.gi_bounces)
        return r

    def uniforms(cam, frame):
        return rt.frame_uniforms(params, cam, W, H, frame, rt.RT_SCENE_HYBRID if args.hybrid else True, nodes.shape[0], tris.shape[0])

    def run_camera(cam, steps, warmup, timed_stage=True):
        """-> dict(ms_per_step, counters summed over the timed frames (this rank), stage times)"""
        # untimed counting pass over the same frame indices (work counters slow the kernels down)
        rc = make_renderer(True)
        for f in range(warmup):
            rc.render_frame(uniforms(cam, f))
        rc.reset_counters()
        for f in range(warmup, warmup + steps):
            rc.render_frame(uniforms(cam, f))
        cnt = rc.counters()
        rc.close()

        ren = make_renderer(False)
        gatherer = None
        if multi:
            # the exchange runs inside the library (rt_comm_init / rt_gather_frame: RCCL behind the C ABI).  If its communicator
            # cannot be brought up on this node the run falls back to the same exchange issued through torch.distributed -- on
            # every rank alike -- and says so in the JSON line.
            ok = torch.ones(1, device="cuda")
            if args.gather == "native":
                try:
                    gatherer = NativeGatherer(ren, gather_every=args.gather_every)
                except Exception as e:   # noqa: BLE001
                    gather_path["error"] = repr(e)[:300]
                    ok.zero_()
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if args.gather != "native" or ok.item() == 0:
                gatherer = FrameGatherer(ren, gather_every=args.gather_every)
                gather_path["path"] = "torch.distributed (RCCL) on the library's device pointers"
        # setup, not a step: every frame lane (3-4 streams with their own ray-queue arenas) allocates on its first frame; do that
        # before the W warm-up steps so that a small W cannot push a multi-GB hipMalloc into the timed region
        B = max(1, min(args.batch, 16))
        setup_u = [uniforms(cam, f) for f in range(5 * B)]
        for b in range(5):
            ren.render_frames(setup_u[b * B:(b + 1) * B])
            if gatherer:
                gatherer.after(B, last=True)
        ren.synchronize()
        ren.reset_accum()
        if gatherer:
            gatherer.frames = 0

        frames_u = [uniforms(cam, f) for f in range(warmup + steps)]   # inputs prepared outside the timed region

        def run_steps(f0, f1):
            """Frames f0..f1-1 in batches of B (the last one shorter): rt_render_frames, then -- tile-parallel -- the RCCL gather of COLOR0
            to rank 0 + un-tiling kernel for the batch's last frame, if a --gather-every boundary was crossed (default: every batch)."""
            f = f0
            while f < f1:
                n = min(B, f1 - f)
                ren.render_frames(frames_u[f:f + n])
                f += n
                if gatherer:
                    gatherer.after(n, last=(f == f1))

        run_steps(0, warmup)
        ren.synchronize()
        torch.cuda.synchronize()
        if timed_stage:
            ren.enable_stage_timing(True)
        if multi:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run_steps(warmup, warmup + steps)
        ren.synchronize()
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
        dt = time.perf_counter() - t0
        stages = ren.stage_times() if timed_stage else None
        traced = ren.traced_rays()
        info = ren.scene_info()
        ren.close()
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        cc = torch.tensor(list(cnt.to_dict().values()), dtype=torch.int64, device="cuda")
        if multi:
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dist.all_reduce(cc, op=dist.ReduceOp.SUM)
        total = rt.RtCounters(*[int(v) for v in cc.tolist()])
        tr = torch.tensor([traced.rays, traced.frames], dtype=torch.int64, device="cuda")
        if multi:
            dist.all_reduce(tr, op=dist.ReduceOp.SUM)
        frames_all = max(int(tr[1].item()) // world, 1)
        traced_per_frame = int(tr[0].item()) * steps // frames_all // steps if traced.frames else 0
        return {"seconds": float(tt.item()), "counters": total, "local_counters": cnt, "stages": stages,
                "traced_per_frame": traced_per_frame, "traced": traced, "scene_info": info, "counted_frames": steps}

    if args.hybrid:
        # the mesh stands among the analytic objects, seen from the reference's default camera (include/app/state.h:129-131)
        M = np.eye(4, dtype=np.float32)
        M[0, 3], M[1, 3], M[2, 3] = -0.1, 1.0, -0.5
        v_, f_ = rt.meshgen.bunny_standin(args.subdiv)
        nodes, tris = rt.build_bvh(rt.gather_triangles(v_, f_, M.T.reshape(-1)))
    closeup = run_camera(scenes.camera("default" if args.hybrid else "closeup"), args.steps, args.warmup)
    # serial stage breakdown (one frame in flight): in the timed run up to 3-4 frames overlap, which stretches every kernel's
    # wall span; this untimed pass shows what each stage costs when it has the GPU to itself
    serial_stages = None
    unbatched_ms = None
    if world == 1 and not args.hybrid:
        old_lanes = os.environ.get("RT_LANES")
        os.environ["RT_LANES"] = "1"
        try:
            r1 = make_renderer(False)
            cam1 = scenes.camera("closeup")
            B1 = max(1, min(args.batch, 16))
            us1 = [uniforms(cam1, f) for f in range(4 * B1)]
            r1.render_frames(us1[:B1])
            r1.render_frames(us1[B1:2 * B1])
            r1.enable_stage_timing(True)
            r1.render_frames(us1[2 * B1:3 * B1])
            r1.render_frames(us1[3 * B1:])
            sst = r1.stage_times()
            serial_stages = {k: v["ms"] / (2 * B1) for k, v in sst["stages"].items()}
            r1.close()
            # frame by frame (one rt_render_frame per step, three frames in flight), for comparison with the batched figure
            if args.no_frame_by_frame:
                raise StopIteration
            del os.environ["RT_LANES"]
            if old_lanes is not None:
                os.environ["RT_LANES"] = old_lanes
            r2 = make_renderer(False)
            us2 = [uniforms(cam1, f) for f in range(48)]
            for u2 in us2[:8]:
                r2.render_frame(u2)
            r2.synchronize()
            t2 = time.perf_counter()
            for u2 in us2[8:]:
                r2.render_frame(u2)
            r2.synchronize()
            unbatched_ms = (time.perf_counter() - t2) / 40 * 1e3
            r2.close()
            os.environ["RT_LANES"] = "1"
        except StopIteration:
            pass
        finally:
            if old_lanes is None:
                os.environ.pop("RT_LANES", None)
            else:
                os.environ["RT_LANES"] = old_lanes
    res = closeup
    rays = res["counters"].rays
    mray = rays / res["seconds"] / 1e6
    ms_per_step = res["seconds"] / args.steps * 1e3

    # roofline of the dominant kernel (stage with the largest device time on this rank)
    roofline = None
    st = res["stages"]
    if st and st["stages"]:
        # dominant = the stage with the largest device time when it has the GPU to itself (with several frames in flight the event spans
        # of the timed region also contain time spent queueing behind other frames' kernels and can rank the stages differently)
        if serial_stages:
            name = max((k for k in serial_stages if k in st["stages"]), key=lambda k: serial_stages[k])
            dom = st["stages"][name]
        else:
            name, dom = max(st["stages"].items(), key=lambda kv: kv[1]["ms"])
        launches = max(int(dom["launches"]), 1)
        avg_ms = dom["ms"] / launches
        overlapped_span_ms = None
        if serial_stages and name in serial_stages:
            # With several frames in flight the event span of a stage includes time its kernel shares the GPU with (or queues
            # behind) other frames' kernels; the one-frame-in-flight pass gives the kernel's own duration, which is what
            # rocprofv3's kernel trace reports (profiles/r02_*kernel_stats_one_frame_in_flight.csv).
            overlapped_span_ms = avg_ms
            avg_ms = serial_stages[name] * args.steps / launches
        lc = res["local_counters"]
        tr = res["traced"]
        info = res["scene_info"]
        frames_tr = max(int(tr.frames), 1)
        per_frame_launches = launches / args.steps
        # (1) HBM roofline, the contract's: ALGORITHMIC bytes of this launch in THIS implementation's layout = what it has to
        # move through HBM at least once: the ray records it reads (32-byte origin/direction + 4-byte tMax, 4 bytes of pixel
        # slot for a primary ray), the results it writes (1 byte per any-hit ray, 8 per closest-hit ray) and the BVH arrays it
        # walks, once (every re-read of a node is served by L1 / L2 / Infinity Cache or is waste).
        rays_k = {"trace_primary": tr.primary, "trace_shadow": tr.shadow + tr.bounceShadow, "trace_gi": tr.bounce}
        rec_k = {"trace_primary": 4 + 8, "trace_shadow": 36 + 1, "trace_gi": 36 + 8}
        bvh_k = {"trace_primary": info.bytesNodes2 + info.bytesPairs, "trace_shadow": info.bytesNodes4 + info.bytesPairs,
                 "trace_gi": info.bytesNodes2 + info.bytesPairs}
        if name in rays_k:
            rays_per_launch = rays_k[name] / frames_tr / per_frame_launches
            bytes_per_launch = rays_per_launch * rec_k[name] + bvh_k[name]
            attribution = ("%d B per ray traced (record in, result out) x %.0f rays + the BVH arrays this kernel walks once (%d B)"
                           % (rec_k[name], rays_per_launch, bvh_k[name]))
        else:
            share = dom["ms"] / max(sum(v["ms"] for v in st["stages"].values()), 1e-9)
            bytes_per_launch = (npix // world) * 36.0 * (1.0 if len(st["stages"]) == 1 else share) / per_frame_launches
            attribution = "36 B per pixel (8 B history read + 28 B of target writes) x this kernel's share of the frame's device time"
        achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9
        # the SURVEY 8d figure (REFERENCE layout: 48 B per nodeFetch / triFetch of the reference's loop for the rays this kernel
        # traces) is kept beside it; it is not a traffic figure of this implementation (4.3x fewer rays are traversed, a node visit
        # is one 64 / 112-byte record instead of 3 x 48 B) and exceeds the HBM peak on cache-resident scenes.
        fetch_rest = lc.nodeFetch + lc.triFetch - lc.fetchPrimary - lc.fetchShadow - lc.fetchAO
        per_kind = {"trace_primary": lc.fetchPrimary, "trace_shadow": lc.fetchShadow + lc.fetchAO, "trace_gi": fetch_rest}
        ref_layout = None
        if name in per_kind:
            ref_bytes = 48.0 * per_kind[name] / res["counted_frames"] / per_frame_launches
            ref_layout = {"algorithmic_bytes_per_launch": ref_bytes, "bytes_per_s_GB": ref_bytes / (avg_ms * 1e-3) / 1e9, "frac": None,
                          "note": "SURVEY 8d units (48 B x the reference loop's nodeFetch + triFetch for these rays, megakernel counting pass); "
                                  "not bytes this implementation moves, so no fraction of a hardware peak is formed from it"}
        # HBM bytes of that kernel per launch from the PMC passes of tools/collect_profiles.sh (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE
        # in separate passes, gfx950 2x fetch correction).  PMC cannot be collected inside this run: the figure is accepted only
        # when the kernel sources it was measured on are the ones running now, else it is dropped.
        traffic, traffic_src = None, None
        kmap = {"trace_shadow": "DualQueueSrc, true", "trace_gi": "QueueSrc, false",
                "trace_primary": "PrimarySrc", "primary": "k_primary", "combine": "k_combine", "gen_direct": "k_gen_direct"}
        tj = ROOT / "profiles" / ("r02_traffic_%s.json" % ("1m" if args.scene == "1m" else "bunny"))
        if world == 1 and tj.exists() and name in kmap and (W, H, SPP) == (1920, 1080, 4):
            tjd = json.load(open(tj))
            if tjd.get("kernel_source_sha256") == kernel_source_sha():
                for k, v in tjd["kernels"].items():
                    if kmap[name] in k:
                        traffic = v["hbm_bytes_per_frame_corrected"] / per_frame_launches
                        traffic_src = {"kind": "profiled_offline", "file": str(tj.relative_to(ROOT)), "measured_at_commit": tjd.get("commit"),
                                       "kernel_source_sha256": tjd.get("kernel_source_sha256")}
            else:
                traffic_src = {"kind": "stale", "file": str(tj.relative_to(ROOT)),
                               "note": "kernel sources changed since the PMC passes; figure dropped"}
        # (2) what binds these kernels when the BVH is cache-resident: the vector L1's gather path (tools/gather.hip +
        # profiles/r02_gather_microbench_pmc.txt).  The traversal kernels count the 16-byte per-lane node / triangle loads they issue.
        l1 = None
        gl = {"trace_primary": tr.gatherLoadsPrimary, "trace_shadow": tr.gatherLoadsShadow, "trace_gi": tr.gatherLoadsBounce}
        if name in gl and tr.frames:
            per_launch = gl[name] / frames_tr / per_frame_launches
            rate = per_launch / (avg_ms * 1e-3) / 1e9
            pk = L1_GATHER_PEAK_G[name]
            l1 = {"unit": "G lane-loads/s (16 B each)", "lane_loads_per_launch": per_launch, "achieved": rate, "peak": pk,
                  "frac": rate / pk, "bytes_per_s_TB": rate * 16 / 1e3,
                  "peak_source": "tools/gather.hip, fully divergent lanes, this kernel's node record shape (profiles/r02_gather_microbench_pmc.txt: "
                                 "~1 TCP cache access per clock and CU).  Lanes of a wave that read the same 16 bytes are merged by the L1, so the "
                                 "kernel's own count over-states its TCP accesses by ~1.4x: profiles/README.md gives the PMC figure"}
        roofline = {"bound": "hbm", "kernel": name, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                    "traffic_frac_of_peak": (traffic / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
                    "avg_launch_ms": avg_ms, "launches": launches, "algorithmic_bytes_per_launch": bytes_per_launch,
                    "attribution": attribution,
                    "bvh_bytes": {"nodes_2wide": info.bytesNodes2, "nodes_4wide": info.bytesNodes4, "triangle_pairs": info.bytesPairs},
                    "cache_resident": bool(info.bytesNodes2 + info.bytesNodes4 + info.bytesPairs < 32 * 2**20),
                    "reference_layout": ref_layout, "l1_gather": l1,
                    "avg_launch_ms_source": "HIP events, one frame in flight" if overlapped_span_ms is not None else "HIP events, timed region",
                    "event_span_ms_with_frames_overlapping": overlapped_span_ms,
                    "note": "hbm frac = compulsory bytes of the launch / its duration / 8 TB/s.  With the BVH resident in L2 / Infinity Cache "
                            "(cache_resident) HBM is not what bounds the kernel -- a low frac is expected; the binding bound is the L1 gather "
                            "path (l1_gather), see DESIGN.md 4.3"}

    out = {
        "metric": "Mray/s @1080p 4spp bunny BVH" if (args.scene, W, H, SPP, args.hybrid) == ("bunny", 1920, 1080, 4, False) else
                  "Mray/s @%dx%d %dspp %s%s" % (W, H, SPP, args.scene, " + analytic scene, %d GI bounces (extension, not in the reference)" % args.gi_bounces if args.hybrid else " BVH"),
        "value": mray, "unit": "Mray/s", "value_traversed": res["traced_per_frame"] * args.steps / res["seconds"] / 1e6,
        "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": ("configs[1]: " if (args.scene, W, H, SPP, args.hybrid) == ("bunny", 1920, 1080, 4, False) else "variant: ")
                               + ("procedural bunny stand-in (icosphere subdiv %d" % args.subdiv if args.scene == "bunny" else "1M-triangle multi-object scene (")
                               + (", %d tris, median-split BVH), %dx%d, %d spp, GI 1 bounce + AO 4, Sky_01 env, close-up camera (-2,1.5,1.0)" % (tris.shape[0], W, H, SPP)
                                  if not args.hybrid else
                                  ", %d tris) inside the reference's analytic scene (floor, diffuse / glass / mirror spheres, light marker) -- EXTENSION mode=hybrid, "
                                  "not expressible in the reference -- %dx%d, %d spp, %d GI bounces + AO 4, Sky_01 env, reference default camera, megakernel" % (tris.shape[0], W, H, SPP, args.gi_bounces)),
                   "pipeline": args.pipeline, "tiles": "16x16 round-robin over ranks" if world > 1 else "single GPU",
                   "gather": gather_path if multi else None,
                   "frames_per_launch_set": max(1, min(args.batch, 16)),
                   "batching": "rt_render_frames: consecutive frames of the static camera (they differ in uFrameIndex and uJitter only) share one set of "
                               "kernel launches; every frame is fully rendered, results are bit-identical to one rt_render_frame per frame",
                   "ms_per_step_frame_by_frame": unbatched_ms,
                   "rays_per_frame": rays // args.steps, "msample_per_s": npix * SPP * args.steps / res["seconds"] / 1e6,
                   "hit_pixels": res["counters"].hitPixels // args.steps,
                   "rays_traversed_per_frame": res["traced_per_frame"],
                   "ray_accounting": "value counts the reference shader's traceBVH/traceBVHShadow calls (SURVEY 8d); "
                                     "identical rays (SPP copies of the primary and AO rays) are traversed once and disk-light "
                                     "shadow rays of exactly zero weight are not traversed: rays_traversed_per_frame"},
        "roofline": roofline,
    }
    if st:
        out["stage_ms_per_frame"] = {k: v["ms"] / args.steps for k, v in st["stages"].items()}
        out["stage_ms_note"] = "HIP-event spans in the timed region; consecutive frames overlap on 3-4 streams, so spans add up to more than ms_per_step"
    if serial_stages:
        out["stage_ms_per_frame_one_frame_in_flight"] = serial_stages

    if not args.no_default_camera and not args.hybrid:
        d = run_camera(scenes.camera("default"), args.steps, args.warmup, timed_stage=False)
        out["default_camera"] = {"value": d["counters"].rays / d["seconds"] / 1e6, "unit": "Mray/s",
                                 "ms_per_step": d["seconds"] / args.steps * 1e3, "rays_per_frame": d["counters"].rays // args.steps,
                                 "hit_pixels": d["counters"].hitPixels // args.steps}

    if rank == 0 and world == 1 and args.cpu_seconds > 0:
        import tempfile
        import oracle as orc
        cores = usable_cores()
        cam = scenes.camera("default" if args.hybrid else "closeup")
        u = uniforms(cam, args.warmup)
        # the oracle's traversal + shade loop compiled -O3 -march=native on THIS host (bit-identical to the -O2 checker
        # build: tests/test_oracle_kat.py), (a) one thread -- the scalar figure the >= 10x target refers to -- (b) all usable cores.
        with tempfile.TemporaryDirectory() as tmp:
            Ln = orc.load(orc.build_native(tmp))
            order = sorted(range(0, H, 16), key=lambda y: abs(y + 8 - H // 2))   # 16-row bands outward from the middle of the frame

            def sample(nthreads, rows):
                rays_cpu, dt, bands = 0, 0.0, 0
                for y in order:
                    for y0 in range(y, min(y + 16, H), rows):
                        t0 = time.perf_counter()
                        _, c1 = orc.render(u, nodes, tris, faces, None, region=(0, y0, W, min(y0 + rows, H)), nthreads=nthreads, L=Ln, gi_bounces=args.gi_bounces)
                        dt += time.perf_counter() - t0
                        rays_cpu += c1.rays
                        bands += 1
                        if dt >= args.cpu_seconds:
                            return rays_cpu, dt, bands
                return rays_cpu, dt, bands

            r1, t1, b1 = sample(1, 2)
            rn, tn, bn = sample(cores, 16)
        what = "oracle (scalar fp32 C++ restatement of shaders/rt, g++ -O3 -march=native -ffp-contract=off), %dx%d/%dspp close-up frame %d" % (W, H, SPP, args.warmup)
        out["cpu_baseline"] = {"value": rn / tn / 1e6, "unit": "Mray/s", "cores": cores, "kind": "port",
                               "sample": f"{what}: {bn} 16-row bands around the middle, {rn} rays in {tn:.1f} s on {cores} threads",
                               "single_thread": {"value": r1 / t1 / 1e6, "unit": "Mray/s", "cores": 1,
                                                 "sample": f"{b1} 2-row bands around the middle, {r1} rays in {t1:.1f} s on 1 thread"},
                               "frame_seconds_estimate": {"threads_1": rays / args.steps / (r1 / t1), "threads_all": rays / args.steps / (rn / tn)},
                               "note": "the CPU traces every reference ray; the GPU pipeline skips duplicates (config.ray_accounting), so compare "
                                       "frame times (frame_seconds_estimate vs ms_per_step), or value_traversed, not the two Mray/s figures"}
    if multi:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()

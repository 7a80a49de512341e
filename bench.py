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

L1_GATHER_PEAK_G = 256 * 2.4   # G lane-loads/s: one divergent 16-byte lane-load per clock and CU (tools/gather.hip)
HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
W, H, SPP = 1920, 1080, 4


def algorithmic_bytes(c, npix):
    """SURVEY.md 8d / BASELINE.md: reference-layout bytes of one frame."""
    return 48 * c.nodeFetch + 48 * c.triFetch + npix * 36 + 12 * c.envLookup


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
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--subdiv", type=int, default=6, help="icosphere subdivisions of the bunny stand-in (6 = 81 920 tris)")
    ap.add_argument("--pipeline", default="auto", choices=["auto", "mega", "wave"])
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the oracle (CPU baseline) sample; 0 = skip")
    ap.add_argument("--no-default-camera", action="store_true")
    # other BASELINE.json configurations, for side measurements (the default line is configs[1], the one `metric` is quoted on)
    ap.add_argument("--size", default="1920x1080", help="framebuffer WxH (configs[3]: 3840x2160)")
    ap.add_argument("--spp", type=int, default=4, help="samples per pixel and frame (configs[2-3]: 16, configs[4]: 64)")
    ap.add_argument("--scene", default="bunny", choices=["bunny", "1m"], help="1m = configs[4]'s 1M-triangle multi-object scene")
    args = ap.parse_args()
    global W, H, SPP
    W, H = (int(v) for v in args.size.lower().split("x"))
    SPP = args.spp

    import torch
    import torch.distributed as dist
    import opengl_raytracing_amd as rt
    from opengl_raytracing_amd.dist_gather import FrameGatherer
    import scenes

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus N>1 must be launched with torch.distributed.run (one process per GPU)")
        args.gpus = world
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    pipeline = {"auto": rt.RT_PIPELINE_AUTO, "mega": rt.RT_PIPELINE_MEGAKERNEL, "wave": rt.RT_PIPELINE_WAVEFRONT}[args.pipeline]
    if args.scene == "1m":
        import numpy as np
        v, fidx = rt.meshgen.million_triangle_scene()
        nodes, tris = rt.build_bvh(rt.gather_triangles(v, fidx, np.eye(4, dtype=np.float32).reshape(-1)))
    else:
        nodes, tris = scenes.bunny_bvh(args.subdiv)
    faces = scenes.env_faces("Sky_01")
    params = rt.default_render_params()
    params.sppPerFrame = SPP
    npix = W * H

    def make_renderer(count):
        # work counters (reference units) come from the reference-shaped megakernel; the timed run uses `pipeline`
        r = rt.Renderer(device=local_rank, rank=rank, world_size=world,
                        pipeline=rt.RT_PIPELINE_MEGAKERNEL if count else pipeline, count_work=count)
        r.upload_bvh(nodes, tris)
        r.upload_env(faces)
        r.resize(W, H)
        return r

    def uniforms(cam, frame):
        return rt.frame_uniforms(params, cam, W, H, frame, True, nodes.shape[0], tris.shape[0])

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
        gatherer = FrameGatherer(ren) if world > 1 else None
        # setup, not a step: every frame lane (3-4 streams with their own ray-queue arenas) allocates on its first frame; do that
        # before the W warm-up steps so that a small W cannot push a multi-GB hipMalloc into the timed region
        for f in range(5):
            ren.render_frame(uniforms(cam, f))
            if gatherer:
                gatherer.gather()
        ren.synchronize()
        ren.reset_accum()

        frames_u = [uniforms(cam, f) for f in range(warmup + steps)]   # inputs prepared outside the timed region

        def step(f):
            ren.render_frame(frames_u[f])
            if gatherer:
                gatherer.gather()      # one RCCL gather of COLOR0 to rank 0 + un-tiling kernel, on the renderer's stream

        for f in range(warmup):
            step(f)
        ren.synchronize()
        torch.cuda.synchronize()
        if timed_stage:
            ren.enable_stage_timing(True)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for f in range(warmup, warmup + steps):
            step(f)
        ren.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        dt = time.perf_counter() - t0
        stages = ren.stage_times() if timed_stage else None
        traced = ren.traced_rays()
        ren.close()
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        cc = torch.tensor(list(cnt.to_dict().values()), dtype=torch.int64, device="cuda")
        if world > 1:
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dist.all_reduce(cc, op=dist.ReduceOp.SUM)
        total = rt.RtCounters(*[int(v) for v in cc.tolist()])
        tr = torch.tensor([traced.rays, traced.frames], dtype=torch.int64, device="cuda")
        if world > 1:
            dist.all_reduce(tr, op=dist.ReduceOp.SUM)
        frames_all = max(int(tr[1].item()) // world, 1)
        traced_per_frame = int(tr[0].item()) * steps // frames_all // steps if traced.frames else 0
        return {"seconds": float(tt.item()), "counters": total, "local_counters": cnt, "stages": stages,
                "traced_per_frame": traced_per_frame, "traced": traced}

    closeup = run_camera(scenes.camera("closeup"), args.steps, args.warmup)
    # serial stage breakdown (one frame in flight): in the timed run up to 3-4 frames overlap, which stretches every kernel's
    # wall span; this untimed pass shows what each stage costs when it has the GPU to itself
    serial_stages = None
    if world == 1:
        old_lanes = os.environ.get("RT_LANES")
        os.environ["RT_LANES"] = "1"
        try:
            r1 = make_renderer(False)
            cam1 = scenes.camera("closeup")
            for f in range(max(args.warmup, 2)):
                r1.render_frame(uniforms(cam1, f))
            r1.enable_stage_timing(True)
            for f in range(max(args.warmup, 2), max(args.warmup, 2) + 8):
                r1.render_frame(uniforms(cam1, f))
            sst = r1.stage_times()
            serial_stages = {k: v["ms"] / 8 for k, v in sst["stages"].items()}
            r1.close()
        finally:
            if old_lanes is None:
                del os.environ["RT_LANES"]
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
        name, dom = max(st["stages"].items(), key=lambda kv: kv[1]["ms"])
        launches = max(int(dom["launches"]), 1)
        avg_ms = dom["ms"] / launches
        overlapped_span_ms = None
        if serial_stages and name in serial_stages:
            # With several frames in flight the event span of a stage includes time its kernel shares the GPU with (or queues
            # behind) other frames' kernels; the one-frame-in-flight pass gives the kernel's own duration, which is what
            # rocprofv3's kernel trace reports (profiles/r01_wavefront_kernel_stats*.csv).
            overlapped_span_ms = avg_ms
            avg_ms = serial_stages[name] * args.steps / launches
        lc = res["local_counters"]
        # algorithmic bytes (reference layout, SURVEY 8d: 48 B per nodeFetch / triFetch) of the rays this kernel traverses,
        # from the per-ray-kind fetch counters of the counting pass; non-traversal stages and the megakernel get the
        # frame's bytes split by device-time share.
        fetch_rest = lc.nodeFetch + lc.triFetch - lc.fetchPrimary - lc.fetchShadow - lc.fetchAO
        per_kind = {"trace_primary": lc.fetchPrimary, "trace_shadow": lc.fetchShadow + lc.fetchAO, "trace_gi": fetch_rest}
        bytes_per_frame = algorithmic_bytes(lc, npix // world) / args.steps
        if name in per_kind:
            bytes_per_launch = 48.0 * per_kind[name] / launches
            attribution = ("48 B x (nodeFetch + triFetch) of the reference's traversal for the rays this kernel traces "
                           "(trace_shadow: traceBVHShadow rays + computeAO rays), megakernel counting pass")
        else:
            share = dom["ms"] / max(sum(v["ms"] for v in st["stages"].values()), 1e-9)   # same overlap factor on both sides
            bytes_per_launch = bytes_per_frame * (1.0 if len(st["stages"]) == 1 else share) * args.steps / launches
            attribution = "frame's reference-layout bytes x this kernel's share of the frame's device time"
        achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9
        # HBM bytes of that kernel per launch from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate
        # passes, gfx950 2x fetch correction; profiles/README.md).  PMC cannot be collected from inside the timed run.
        traffic, traffic_src = None, None
        tj = ROOT / "profiles" / "r01_wavefront_traffic.json"
        kmap = {"trace_shadow": "DualQueueSrc, true", "trace_gi": "QueueSrc, false",
                "trace_primary": "PrimarySrc", "primary": "k_primary", "combine": "k_combine", "gen_direct": "k_gen_direct"}
        if world == 1 and tj.exists() and name in kmap:
            for k, v in json.load(open(tj))["kernels"].items():
                if kmap[name] in k:
                    per_frame = v["hbm_bytes_per_frame_corrected"]
                    traffic = per_frame * args.steps / launches
                    traffic_src = "profiles/r01_wavefront_traffic.json"
        # The bound these kernels really run against: divergent per-lane gathers go through a CU's L1 at one 16-byte lane-load per
        # clock (tools/gather.hip, profiles/r01_gather_microbench.txt: 591 G lane-loads/s measured chip-wide, 256 CUs x 2.4 GHz =
        # 614 G/s in the model).  The traversal kernels count the node / triangle gather loads they issue (RtTracedRays.gatherLoads*).
        l1 = None
        tr = res["traced"]
        gl = {"trace_primary": tr.gatherLoadsPrimary, "trace_shadow": tr.gatherLoadsShadow, "trace_gi": tr.gatherLoadsBounce}
        if name in gl and tr.frames:
            per_launch = gl[name] / tr.frames * args.steps / launches
            rate = per_launch / (avg_ms * 1e-3) / 1e9
            l1 = {"unit": "G lane-loads/s (16 B each)", "lane_loads_per_launch": per_launch, "achieved": rate, "peak": L1_GATHER_PEAK_G,
                  "frac": rate / L1_GATHER_PEAK_G, "bytes_per_s_TB": rate * 16 / 1e3,
                  "peak_source": "256 CUs x 2.4 GHz x 1 lane-load/clk; tools/gather.hip measures 591 G/s for 64 lanes x 4 x dwordx4 from a 1 MB table"}
        roofline = {"bound": "hbm", "kernel": name, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                    "avg_launch_ms": avg_ms, "launches": launches, "algorithmic_bytes_per_launch": bytes_per_launch,
                    "attribution": attribution, "l1_gather": l1,
                    "avg_launch_ms_source": "HIP events, one frame in flight" if overlapped_span_ms is not None else "HIP events, timed region",
                    "event_span_ms_with_frames_overlapping": overlapped_span_ms,
                    "note": "BVH (1 MB nodes + 3.9 MB tris) is L2/Infinity-Cache resident and the reference layout fetches 3x48 B per node visit, "
                            "so algorithmic bytes/s exceed the HBM peak (frac > 1) while measured HBM traffic is ~8% of peak: the kernel is bound by "
                            "the L1 gather rate (l1_gather), see DESIGN.md 4.3 and profiles/README.md"}

    out = {
        "metric": "Mray/s @1080p 4spp bunny BVH" if (args.scene, W, H, SPP) == ("bunny", 1920, 1080, 4) else "Mray/s @%dx%d %dspp %s BVH" % (W, H, SPP, args.scene),
        "value": mray, "unit": "Mray/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": ("configs[1]: " if (args.scene, W, H, SPP) == ("bunny", 1920, 1080, 4) else "variant: ")
                               + ("procedural bunny stand-in (icosphere subdiv %d" % args.subdiv if args.scene == "bunny" else "1M-triangle multi-object scene (")
                               + ", %d tris, median-split BVH), %dx%d, %d spp, GI 1 bounce + AO 4, Sky_01 env, close-up camera (-2,1.5,1.0)" % (tris.shape[0], W, H, SPP),
                   "pipeline": args.pipeline, "tiles": "16x16 round-robin over ranks" if world > 1 else "single GPU",
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

    if not args.no_default_camera:
        d = run_camera(scenes.camera("default"), args.steps, args.warmup, timed_stage=False)
        out["default_camera"] = {"value": d["counters"].rays / d["seconds"] / 1e6, "unit": "Mray/s",
                                 "ms_per_step": d["seconds"] / args.steps * 1e3, "rays_per_frame": d["counters"].rays // args.steps,
                                 "hit_pixels": d["counters"].hitPixels // args.steps}

    if rank == 0 and world == 1 and args.cpu_seconds > 0:
        import oracle as orc
        cores = usable_cores()
        cam = scenes.camera("closeup")
        u = uniforms(cam, args.warmup)
        # bands of 16 rows outward from the middle of the frame until the time budget is used
        order = sorted(range(0, H, 16), key=lambda y: abs(y + 8 - H // 2))
        rays_cpu, dt, bands = 0, 0.0, 0
        for y in order:
            t0 = time.perf_counter()
            _, c1 = orc.render(u, nodes, tris, faces, None, region=(0, y, W, min(y + 16, H)), nthreads=cores)
            dt += time.perf_counter() - t0
            rays_cpu += c1.rays
            bands += 1
            if dt >= args.cpu_seconds:
                break
        out["cpu_baseline"] = {"value": rays_cpu / dt / 1e6, "unit": "Mray/s", "cores": cores, "kind": "port",
                               "sample": f"oracle (scalar fp32 C++ restatement of shaders/rt, g++ -O2), {bands} 16-row bands around the "
                                         f"middle of the same 1080p/4spp close-up frame {args.warmup}: {rays_cpu} rays in {dt:.1f} s "
                                         f"on {cores} threads"}
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()

"""Render a few frames of the bench workload (no counting pass, no CPU baseline) -- the target of rocprofv3 runs."""
import sys, argparse
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
import opengl_raytracing_amd as rt, scenes

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=4)
ap.add_argument("--camera", default="closeup")
ap.add_argument("--pipeline", default="wave")
ap.add_argument("--spp", type=int, default=4)
ap.add_argument("--subdiv", type=int, default=6)
ap.add_argument("--size", default="1920x1080")
ap.add_argument("--scene", default="bunny")
ap.add_argument("--objects", type=int, default=16, help="--scene 1m: number of displaced spheres (62 500 triangles each; 16 = the 1 M-triangle scene)")
ap.add_argument("--world", type=int, default=1)
ap.add_argument("--rank", type=int, default=0)
ap.add_argument("--batch", type=int, default=1, help="frames per rt_render_frames call (bench.py's default is 8)")
a = ap.parse_args()
W, H = map(int, a.size.split("x"))
import time
if a.scene == "1m":
    t0 = time.time(); v, f = rt.meshgen.million_triangle_scene(a.objects); tris9 = rt.gather_triangles(v, f, np.eye(4, dtype=np.float32).T.reshape(-1)); nodes, tris = rt.build_bvh(tris9)
    print('1M scene', tris.shape, nodes.shape, 'built in', round(time.time() - t0, 1), 's')
else:
    nodes, tris = scenes.bunny_bvh(a.subdiv)
faces = scenes.env_faces("Sky_01")
p = rt.default_render_params(); p.sppPerFrame = a.spp
cam = scenes.camera(a.camera, aspect=W / H)
pipe = {"wave": rt.RT_PIPELINE_WAVEFRONT, "mega": rt.RT_PIPELINE_MEGAKERNEL}[a.pipeline]
with rt.Renderer(pipeline=pipe, rank=a.rank, world_size=a.world) as r:
    r.upload_bvh(nodes, tris); r.upload_env(faces); r.resize(W, H)
    for f in range(3):
        r.render_frame(rt.frame_uniforms(p, cam, W, H, f, True, nodes.shape[0], tris.shape[0]))
    r.synchronize()
    r.enable_stage_timing(True)
    us = [rt.frame_uniforms(p, cam, W, H, f, True, nodes.shape[0], tris.shape[0]) for f in range(3, 3 + a.frames)]
    for f in range(0, a.frames, max(a.batch, 1)):
        r.render_frames(us[f:f + max(a.batch, 1)])
    r.synchronize()
    st = r.stage_times()
    d = {k: round(v["ms"] / a.frames, 3) for k, v in st["stages"].items()}
    d["total"] = round(sum(d.values()), 3)
    print(d, r.traced_rays(True).to_dict())

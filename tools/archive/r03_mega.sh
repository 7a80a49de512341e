#!/bin/bash
# megakernel timings: analytic scene 1080p, BVH scene (reference-shaped baseline), hybrid on the megakernel
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import opengl_raytracing_amd as rt, scenes
W, H = 1920, 1080
p = rt.default_render_params()
cam = scenes.camera("default")
with rt.Renderer() as r:
    r.upload_env(scenes.env_faces("Sky_01")); r.resize(W, H)
    us = [rt.frame_uniforms(p, cam, W, H, f, False) for f in range(24)]
    for u in us[:4]: r.render_frame(u)
    r.synchronize(); t = time.perf_counter()
    for u in us[4:]: r.render_frame(u)
    r.synchronize(); print("analytic scene 1080p 1 spp: %.3f ms/frame" % ((time.perf_counter() - t) / 20 * 1e3))
nodes, tris = scenes.bunny_bvh(6); p.sppPerFrame = 4; cam = scenes.camera("closeup")
with rt.Renderer(pipeline=rt.RT_PIPELINE_MEGAKERNEL) as r:
    r.upload_bvh(nodes, tris); r.upload_env(scenes.env_faces("Sky_01")); r.resize(W, H)
    us = [rt.frame_uniforms(p, cam, W, H, f, True, nodes.shape[0], tris.shape[0]) for f in range(10)]
    for u in us[:2]: r.render_frame(u)
    r.synchronize(); t = time.perf_counter()
    for u in us[2:]: r.render_frame(u)
    r.synchronize(); print("BVH scene on the megakernel 1080p 4 spp: %.2f ms/frame" % ((time.perf_counter() - t) / 8 * 1e3))
PY
timeout -k 10 600 python3 bench.py --hybrid --spp 4 --gi-bounces 1 --steps 6 --warmup 2 --cpu-seconds 0 --no-default-camera --no-frame-by-frame --pipeline mega 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('hybrid 4spp/1b on the megakernel ms/frame %.2f' % d['ms_per_step'])"

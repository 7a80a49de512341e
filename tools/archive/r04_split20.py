"""The driver's short run: 20 timed frames after 5 warm-up frames, dealt over batches in different ways (pipeline fill and drain are ~5 % of such a run)."""
import sys, time, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, R + "/tests")
import opengl_raytracing_amd as rt, scenes
W, H = 1920, 1080
nodes, tris = scenes.bunny_bvh(6); faces = scenes.env_faces("Sky_01"); p = rt.default_render_params(); p.sppPerFrame = 4; cam = scenes.camera("closeup")
us = [rt.frame_uniforms(p, cam, W, H, f, True, nodes.shape[0], tris.shape[0]) for f in range(80)]
splits = [[7, 7, 6], [8, 8, 4], [8, 7, 5], [8, 6, 6], [10, 10], [6, 5, 5, 4], [5, 5, 5, 5], [8, 8, 2, 2], [4, 8, 8], [7, 7, 6]]
r = rt.Renderer(); r.upload_bvh(nodes, tris); r.upload_env(faces); r.resize(W, H)
for b in range(5): r.render_frames(us[b * 8:(b + 1) * 8])       # set-up: every lane allocates
r.synchronize()
for sp in splits:
    res = []
    for rep in range(3):
        r.reset_accum()
        r.render_frames(us[0:5]); r.synchronize()
        t = time.perf_counter(); f = 5
        for n in sp:
            r.render_frames(us[f:f + n]); f += n
        r.synchronize(); res.append((time.perf_counter() - t) / 20 * 1e3)
    print(sp, " ".join("%.3f" % v for v in res), "ms/frame")

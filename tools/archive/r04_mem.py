import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, R + "/tests")
import torch
import opengl_raytracing_amd as rt, scenes
free0, total = torch.cuda.mem_get_info()
nodes, tris = scenes.bunny_bvh(6); faces = scenes.env_faces("Sky_01"); p = rt.default_render_params(); p.sppPerFrame = 4; cam = scenes.camera("closeup")
W, H, K = 1920, 1080, 8
r = rt.Renderer(); r.upload_bvh(nodes, tris); r.upload_env(faces); r.resize(W, H)
us = [rt.frame_uniforms(p, cam, W, H, f, True, nodes.shape[0], tris.shape[0]) for f in range(12 * K)]
for b in range(4): r.render_frames(us[b * K:(b + 1) * K])
r.synchronize(); t = time.perf_counter()
for b in range(4, 12): r.render_frames(us[b * K:(b + 1) * K])
r.synchronize(); dt = (time.perf_counter() - t) / (8 * K) * 1e3
free1, _ = torch.cuda.mem_get_info()
print("lib", rt.LIB_PATH, "ms/frame %.3f" % dt, "device memory taken by the context: %.2f GB (free before %.2f, after %.2f, total %.2f)" % ((free0 - free1) / 1e9, free0 / 1e9, free1 / 1e9, total / 1e9))

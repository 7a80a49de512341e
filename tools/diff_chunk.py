import sys, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import opengl_raytracing_amd as rt, oracle as orc, scenes
W, H = 200, 120
nodes, tris = scenes.bunny_bvh(4); faces = scenes.tiny_env(16)
import os
p = rt.default_render_params(); p.sppPerFrame = 2
for kv in os.environ.get('TOG','').split(','):
    if '=' in kv:
        k,v=kv.split('='); setattr(p,k,int(v))
cam = scenes.camera("closeup", aspect=W / H)
with rt.Renderer(pipeline=rt.RT_PIPELINE_WAVEFRONT) as r:
    r.upload_bvh(nodes, tris); r.upload_env(faces); r.resize(W, H)
    u = rt.frame_uniforms(p, cam, W, H, 0, True, nodes.shape[0], tris.shape[0])
    r.render_frame(u)
    got = r.read_all()
    want, cnt = orc.render(u, nodes, tris, faces, None)
    print('hit pixels', cnt.hitPixels, r.traced_rays().to_dict())
    for name, g, w in zip(("color","motion","gpos","gnrm"), got, want):
        d = np.argwhere((g != w).any(-1))
        print(name, 'ndiff px', len(d))
        if len(d):
            ys, xs = d[:,0], d[:,1]
            print('  y range', ys.min(), ys.max(), 'x range', xs.min(), xs.max(), 'sample', d[:5].tolist())
            # tile ids of differing pixels
            t = (ys//16)*((W+15)//16) + xs//16
            print('  tiles', np.unique(t)[:40])
    print('got  color', orc.half_to_float(got[0][0,0]), 'gpos', orc.half_to_float(got[2][0,0]), 'gnrm', orc.half_to_float(got[3][0,0]))
    print('want color', orc.half_to_float(want[0][0,0]), 'gpos', orc.half_to_float(want[2][0,0]), 'gnrm', orc.half_to_float(want[3][0,0]))

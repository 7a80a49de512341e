import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np
import opengl_raytracing_amd as rt, oracle as orc, scenes
W=H=256
ren=rt.Renderer(count_work=True, pipeline=rt.RT_PIPELINE_MEGAKERNEL)
ren.upload_env(None); ren.resize(W,H)
p=rt.default_render_params(); p.enableEnvMap=0
cam=scenes.camera("default",aspect=1.0)
for variant in range(6,10):
    q=p.copy()
    if variant==1: q.enableGI=0
    if variant==2: q.enableAO=0
    if variant==3: q.enableGI=0; q.enableAO=0
    if variant==4: q.enableGI=0; q.enableAO=0; q.sunEnabled=0; q.pointLightEnabled=0
    if variant==6: q.enableGI=0; q.enableAO=0; q.pointLightEnabled=0
    if variant==7: q.enableGI=0; q.enableAO=0; q.sunEnabled=0
    if variant==8: q.enableGI=0; q.enableAO=0; q.sunEnabled=0; q.enableTAA=0; q.skyEnabled=0
    if variant==9: q.enableGI=0; q.enableAO=0; q.pointLightEnabled=0; q.skyEnabled=0
    if variant==5: q.enableGI=0; q.enableAO=0; q.sunEnabled=0; q.pointLightEnabled=0; q.skyEnabled=0
    ren.reset_accum()
    u=rt.frame_uniforms(q,cam,W,H,0,False)
    ren.render_frame(u); got=ren.read_all(); want,_=orc.render(u)
    d=np.argwhere(got[0]!=want[0])
    print('variant',variant,'ndiff',len(d))
    for (y,x,c) in d[:10]:
        print('  px',x,y,'ch',c,'got',got[0][y,x], 'want',want[0][y,x], 'gpos', orc.half_to_float(want[2][y,x]), 'nrm', orc.half_to_float(want[3][y,x]))

import sys,time,os; R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,R); sys.path.insert(0,R+"/tests")
import opengl_raytracing_amd as rt, scenes
W,H=1920,1080; world=int(sys.argv[1]) if len(sys.argv)>1 else 1
nodes,tris=scenes.bunny_bvh(6); faces=scenes.env_faces("Sky_01"); p=rt.default_render_params(); p.sppPerFrame=4; cam=scenes.camera(sys.argv[2] if len(sys.argv)>2 else "closeup")
r=rt.Renderer(rank=0,world_size=world); r.upload_bvh(nodes,tris); r.upload_env(faces); r.resize(W,H)
N=60
us=[rt.frame_uniforms(p,cam,W,H,f,True,nodes.shape[0],tris.shape[0]) for f in range(N+5)]
for f in range(5): r.render_frame(us[f])
r.synchronize(); t=time.perf_counter()
for f in range(5,N+5): r.render_frame(us[f])
t_issue=time.perf_counter()-t
r.synchronize(); print("host issue", round(t_issue/N*1e3,3), "ms/frame;", end=" "); print(round((time.perf_counter()-t)/N*1e3,3),"ms/frame")

import sys,time,os; R=os.environ.get("GRAFT_REPO_ROOT","/root/repo"); sys.path.insert(0,R); sys.path.insert(0,R+"/tests")
import opengl_raytracing_amd as rt, scenes
W,H=int(sys.argv[2]),int(sys.argv[3]); world=int(sys.argv[1])
nodes,tris=scenes.bunny_bvh(6); faces=scenes.env_faces("Sky_01"); p=rt.default_render_params(); p.sppPerFrame=4; cam=scenes.camera("closeup")
r=rt.Renderer(rank=0,world_size=world); r.upload_bvh(nodes,tris); r.upload_env(faces); r.resize(W,H)
N=40
us=[rt.frame_uniforms(p,cam,W,H,f,True,nodes.shape[0],tris.shape[0]) for f in range(N+5)]
for f in range(5): r.render_frame(us[f])
r.synchronize(); t=time.perf_counter()
for f in range(5,N+5): r.render_frame(us[f])
r.synchronize(); print(world, W, H, round((time.perf_counter()-t)/N*1e3,3),"ms/frame")

"""wall ms/frame of the bench frame with rt_render_frames in batches of K:  wall_batch.py <world> <K> [size] [rank]   (one rank of `world`, emulated on this GPU)"""
import sys,time,os; R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,R); sys.path.insert(0,R+"/tests")
import opengl_raytracing_amd as rt, scenes
world=int(sys.argv[1]); K=int(sys.argv[2]); W,H=(int(v) for v in (sys.argv[3] if len(sys.argv)>3 else "1920x1080").split("x"))
nodes,tris=scenes.bunny_bvh(6); faces=scenes.env_faces("Sky_01"); p=rt.default_render_params(); p.sppPerFrame=4; cam=scenes.camera("closeup")
rank=int(sys.argv[4]) if len(sys.argv)>4 else 0
r=rt.Renderer(rank=rank,world_size=world); r.upload_bvh(nodes,tris); r.upload_env(faces); r.resize(W,H)
N=64
us=[rt.frame_uniforms(p,cam,W,H,f,True,nodes.shape[0],tris.shape[0]) for f in range(N+4*K+K)]
for b in range(4): r.render_frames(us[b*K:(b+1)*K])
r.synchronize(); t=time.perf_counter(); f=4*K
while f < 4*K+N:
    r.render_frames(us[f:f+K]); f+=K
r.synchronize(); print("world",world,"rank",rank,"K",K,round((time.perf_counter()-t)/N*1e3,3),"ms/frame")

"""CPU prototype (no GPU): does a near-first child order shorten any-hit traversals?  Only occluded rays can gain.  python tools/r03_anyhit_order_proto.py [bunny|1m] [rays]"""
import sys, numpy as np, time
import os; R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, R + "/tests")
import opengl_raytracing_amd as rt, scenes
scene = sys.argv[1] if len(sys.argv) > 1 else "bunny"
if scene == "1m":
    v, f = rt.meshgen.million_triangle_scene()
    nodes, tris = rt.build_bvh(rt.gather_triangles(v, f, np.eye(4, dtype=np.float32).reshape(-1)))
else:
    nodes, tris = scenes.bunny_bvh(6)
left = (nodes[:, 3] + 0.5).astype(int); right = (nodes[:, 7] + 0.5).astype(int)
first = (nodes[:, 8] + 0.5).astype(int); count = (nodes[:, 9] + 0.5).astype(int)
bmin = nodes[:, 0:3].astype(np.float64); bmax = nodes[:, 4:7].astype(np.float64)
isleaf = count > 0
T = tris.reshape(-1, 12).astype(np.float64)
sys.setrecursionlimit(100000)
w4 = []
def make(b):
    me = len(w4); w4.append(None)
    kids = []
    for ch in (left[b], right[b]):
        if isleaf[ch]: kids.append(ch)
        else: kids += [left[ch], right[ch]]
    out = []
    for k in kids:
        if isleaf[k]: out.append(('L', k, bmin[k], bmax[k]))
        else: out.append(('I', make(k), bmin[k], bmax[k]))
    w4[me] = out
    return me
make(0)
def leaf_hit(idx, ro, rd, tmax):
    for t in T[first[idx]:first[idx] + count[idx]]:
        v0, e1, e2 = t[0:3], t[4:7], t[8:11]
        p = np.cross(rd, e2); det = e1 @ p
        if abs(det) < 1e-8: continue
        inv = 1 / det; tv = ro - v0; u = (tv @ p) * inv
        if u < 0 or u > 1: continue
        q = np.cross(tv, e1); vv = (rd @ q) * inv
        if vv < 0 or u + vv > 1: continue
        tt = (e2 @ q) * inv
        if tt < 1e-4 or tt > tmax: continue
        return True
    return False
def traverse(ro, rd, tmax, mode):
    inv = 1.0 / rd
    visits = leafv = 0
    st = [0]; pend = []
    while st or pend:
        if not st:   # leaf phase
            l = pend.pop(); leafv += 1
            if leaf_hit(l, ro, rd, tmax): return visits, leafv, 1
            continue
        n = st.pop(); visits += 1
        cand = []
        for kind, idx, lo, hi in w4[n]:
            t0 = (lo - ro) * inv; t1 = (hi - ro) * inv
            tn = max(np.minimum(t0, t1).max(), 0.0); tf = np.maximum(t0, t1).min()
            if tf >= tn and tn <= tmax: cand.append((tn, kind, idx))
        if mode == "near": cand.sort(key=lambda c: -c[0])      # push far first -> near popped first
        for tn, kind, idx in cand:
            if kind == 'L':
                if mode == "near":
                    leafv += 1
                    if leaf_hit(idx, ro, rd, tmax): return visits, leafv, 1   # leaf tested right away in near-first mode
                else: pend.append(idx)
            else: st.append(idx)
        if mode == "fixed" and len(pend) >= 1 and not st: pass
    return visits, leafv, 0
rng = np.random.default_rng(1)
def make_rays(n):
    out = []
    while len(out) < n:
        t = T[rng.integers(0, T.shape[0])]
        v0, e1, e2 = t[0:3], t[4:7], t[8:11]
        p = v0 + (e1 + e2) / 3
        nrm = np.cross(e1, e2); ln = np.linalg.norm(nrm)
        if ln == 0: continue
        nrm /= ln
        u1, u2 = rng.random(), rng.random()
        r = np.sqrt(u2); phi = 2 * np.pi * u1
        up = np.array([0, 1.0, 0]) if abs(nrm[1]) < 0.99 else np.array([1.0, 0, 0])
        tx = np.cross(up, nrm); tx /= np.linalg.norm(tx); bx = np.cross(nrm, tx)
        d = r * np.cos(phi) * tx + r * np.sin(phi) * bx + np.sqrt(max(0, 1 - u2)) * nrm
        d /= np.linalg.norm(d)
        out.append((p + nrm * 0.002, d, 0.8))
    return out
rays = make_rays(int(sys.argv[2]) if len(sys.argv) > 2 else 500)
for mode in ("fixed", "near"):
    r = np.array([traverse(*ray, mode) for ray in rays], dtype=np.float64)
    occ = r[:, 2] > 0
    print(f"{mode:6s}: occluded {occ.mean():.2f} | all rays: inner {r[:,0].mean():.1f} leaf {r[:,1].mean():.1f} | occluded rays: inner {r[occ,0].mean():.1f} leaf {r[occ,1].mean():.1f} | free rays: inner {r[~occ,0].mean():.1f} leaf {r[~occ,1].mean():.1f}")

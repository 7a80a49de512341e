"""CPU prototype (no GPU): what a PACKET of the four AO rays of one hit (same origin hp + N*bias, cosine directions, tMax 0.8) costs on the library's
4-wide any-hit tree, against the four rays traced one by one: 4-wide node visits and leaf visits of the union (a node is visited once if any ray of
the packet enters it) vs. the sum over the rays.  Full traversal (99.8 % of AO rays are unoccluded).   python tools/r04_packet_proto.py [hits]"""
import sys, numpy as np, time, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, R + "/tests")
import opengl_raytracing_amd as rt, scenes
nodes, tris = scenes.bunny_bvh(6)
left = (nodes[:, 3] + 0.5).astype(int); right = (nodes[:, 7] + 0.5).astype(int); count = (nodes[:, 9] + 0.5).astype(int)
bmin = nodes[:, 0:3].astype(np.float64); bmax = nodes[:, 4:7].astype(np.float64); isleaf = count > 0
w4 = []
def make(b):
    me = len(w4); w4.append(None); kids = []
    for ch in (left[b], right[b]):
        if isleaf[ch]: kids.append(ch)
        else: kids += [left[ch], right[ch]]
    w4[me] = [('L', k, bmin[k], bmax[k]) if isleaf[k] else ('I', make(k), bmin[k], bmax[k]) for k in kids]
    return me
sys.setrecursionlimit(100000); make(0)
def hits_box(ro, inv, lo, hi, tmax):
    t0 = (lo - ro) * inv; t1 = (hi - ro) * inv
    tn = max(np.minimum(t0, t1).max(), 0.0); tf = np.maximum(t0, t1).min()
    return tf >= tn and tn <= tmax
def trav(ro, dirs, tmax):
    """-> per-ray (node visits, leaf visits) summed, packet (node visits, leaf visits), slab tests per-ray / packet"""
    invs = [1.0 / d for d in dirs]
    single_n = single_l = 0
    for inv in invs:
        st = [0]
        while st:
            n = st.pop(); single_n += 1
            for kind, idx, lo, hi in w4[n]:
                if hits_box(ro, inv, lo, hi, tmax):
                    if kind == 'L': single_l += 1
                    else: st.append(idx)
    pk_n = pk_l = 0
    st = [(0, (1 << len(dirs)) - 1)]
    while st:
        n, mask = st.pop(); pk_n += 1
        for kind, idx, lo, hi in w4[n]:
            m = 0
            for r, inv in enumerate(invs):
                if (mask >> r) & 1 and hits_box(ro, inv, lo, hi, tmax): m |= 1 << r
            if m:
                if kind == 'L': pk_l += 1
                else: st.append((idx, m))
    return single_n, single_l, pk_n, pk_l
rng = np.random.default_rng(1); T = tris.reshape(-1, 12)
def hemi(nrm):
    u1, u2 = rng.random(), rng.random(); r = np.sqrt(u2); phi = 2 * np.pi * u1
    up = np.array([0, 1.0, 0]) if abs(nrm[1]) < 0.99 else np.array([1.0, 0, 0])
    tx = np.cross(up, nrm); tx /= np.linalg.norm(tx); bx = np.cross(nrm, tx)
    d = r * np.cos(phi) * tx + r * np.sin(phi) * bx + np.sqrt(max(0, 1 - u2)) * nrm
    return d / np.linalg.norm(d)
acc = np.zeros(4); n = int(sys.argv[1]) if len(sys.argv) > 1 else 300; t0 = time.time()
for _ in range(n):
    t = T[rng.integers(0, T.shape[0])]
    v0, e1, e2 = t[0:3].astype(np.float64), t[4:7].astype(np.float64), t[8:11].astype(np.float64)
    nrm = np.cross(e1, e2); nrm /= np.linalg.norm(nrm)
    acc += trav(v0 + (e1 + e2) / 3 + nrm * 0.002, [hemi(nrm) for _ in range(4)], 0.8)
acc /= n
print(f"{n} hits x 4 AO rays: one by one: {acc[0]:.1f} node visits + {acc[1]:.1f} leaf visits per hit ({acc[0]/4:.1f} + {acc[1]/4:.1f} per ray); "
      f"as a packet: {acc[2]:.1f} node visits + {acc[3]:.1f} leaf visits per hit -> x{acc[2]/acc[0]:.2f} node records, x{acc[3]/acc[1]:.2f} leaf records fetched ({time.time()-t0:.0f}s)")

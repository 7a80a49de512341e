"""CPU prototype (no GPU): node / leaf visits of any-hit rays on the 4-wide tree the library builds from the reference's median-split BVH vs a binned-SAH tree
over the SAME leaves -- AO-like rays (origin on the surface, cosine direction, tMax 0.8), full traversal.  python tools/r03_tree_proto.py [bunny|1m] [rays]"""
import sys, numpy as np, time
import os; R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, R + "/tests")
import opengl_raytracing_amd as rt, scenes
scene = sys.argv[1] if len(sys.argv) > 1 else "bunny"
if scene == "1m":
    v, f = rt.meshgen.million_triangle_scene()
    nodes, tris = rt.build_bvh(rt.gather_triangles(v, f, np.eye(4, dtype=np.float32).reshape(-1)))
else:
    nodes, tris = scenes.bunny_bvh(6)
N = nodes.shape[0]
left = (nodes[:, 3] + 0.5).astype(int); right = (nodes[:, 7] + 0.5).astype(int)
first = (nodes[:, 8] + 0.5).astype(int); count = (nodes[:, 9] + 0.5).astype(int)
bmin = nodes[:, 0:3].astype(np.float64); bmax = nodes[:, 4:7].astype(np.float64)
isleaf = count > 0
print(scene, "nodes", N, "leaves", isleaf.sum(), "tris", tris.shape[0])

def sa(lo, hi):
    e = np.maximum(hi - lo, 0); return 2 * (e[0] * e[1] + e[1] * e[2] + e[2] * e[0])

# ---- generic 4-wide tree: list of nodes, each = list of (childkind, idx) with kind 'L' leaf (ref node idx) or 'I' inner (index into w4)
def collapse_median():
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
    sys.setrecursionlimit(100000)
    make(0)
    return w4

def build_sah(width=4, bins=16):
    leaves = np.nonzero(isleaf)[0]
    lo = bmin[leaves]; hi = bmax[leaves]; cen = 0.5 * (lo + hi); wt = count[leaves].astype(np.float64)
    # binary SAH tree: nodes as dict
    B = []   # (lo, hi, leftchild, rightchild, leafidx)
    def rec(ids, depth):
        blo = lo[ids].min(0); bhi = hi[ids].max(0)
        me = len(B); B.append(None)
        if len(ids) == 1:
            B[me] = (blo, bhi, -1, -1, leaves[ids[0]]); return me
        clo = cen[ids].min(0); chi = cen[ids].max(0)
        best = (np.inf, -1, -1)
        for ax in range(3):
            ext = chi[ax] - clo[ax]
            if ext <= 0: continue
            bi = np.minimum(((cen[ids, ax] - clo[ax]) / ext * bins).astype(int), bins - 1)
            cnt = np.bincount(bi, weights=wt[ids], minlength=bins)
            blo_b = np.full((bins, 3), np.inf); bhi_b = np.full((bins, 3), -np.inf)
            np.minimum.at(blo_b, bi, lo[ids]); np.maximum.at(bhi_b, bi, hi[ids])
            # sweep
            l_lo = np.minimum.accumulate(blo_b, 0); l_hi = np.maximum.accumulate(bhi_b, 0); l_n = np.cumsum(cnt)
            r_lo = np.minimum.accumulate(blo_b[::-1], 0)[::-1]; r_hi = np.maximum.accumulate(bhi_b[::-1], 0)[::-1]; r_n = np.cumsum(cnt[::-1])[::-1]
            for s in range(bins - 1):
                if l_n[s] == 0 or r_n[s + 1] == 0: continue
                c = sa(l_lo[s], l_hi[s]) * l_n[s] + sa(r_lo[s + 1], r_hi[s + 1]) * r_n[s + 1]
                if c < best[0]: best = (c, ax, s)
        if best[1] < 0 or depth > 40:
            ax = int(np.argmax(chi - clo)); order = np.argsort(cen[ids, ax], kind="stable"); h = len(ids) // 2
            L, R = ids[order[:h]], ids[order[h:]]
        else:
            ax, s = best[1], best[2]
            ext = chi[ax] - clo[ax]
            bi = np.minimum(((cen[ids, ax] - clo[ax]) / ext * bins).astype(int), bins - 1)
            L, R = ids[bi <= s], ids[bi > s]
        l = rec(L, depth + 1); r = rec(R, depth + 1)
        B[me] = (blo, bhi, l, r, -1)
        return me
    sys.setrecursionlimit(100000)
    rec(np.arange(len(leaves)), 0)
    # collapse to width
    w4 = []
    def make(b):
        me = len(w4); w4.append(None)
        kids = [B[b][2], B[b][3]]
        while len(kids) < width:
            cand = [(sa(B[k][0], B[k][1]), i) for i, k in enumerate(kids) if B[k][4] < 0]
            if not cand: break
            _, i = max(cand)
            k = kids.pop(i); kids += [B[k][2], B[k][3]]
        out = []
        for k in kids:
            if B[k][4] >= 0: out.append(('L', B[k][4], B[k][0], B[k][1]))
            else: out.append(('I', make(k), B[k][0], B[k][1]))
        w4[me] = out
        return me
    make(0)
    return w4

def depth_of(w4):
    d = [0] * len(w4)
    best = 0
    st = [(0, 1)]
    while st:
        n, dd = st.pop(); best = max(best, dd)
        for k in w4[n]:
            if k[0] == 'I': st.append((k[1], dd + 1))
    return best

def traverse(w4, ro, rd, tmax):
    inv = 1.0 / rd
    visits = leafv = tests = 0
    st = [0]
    while st:
        n = st.pop(); visits += 1
        for kind, idx, lo, hi in w4[n]:
            t0 = (lo - ro) * inv; t1 = (hi - ro) * inv
            tn = max(np.minimum(t0, t1).max(), 0.0); tf = np.maximum(t0, t1).min()
            if tf >= tn and tn <= tmax:
                if kind == 'L': leafv += 1; tests += count[idx]
                else: st.append(idx)
    return visits, leafv, tests

rng = np.random.default_rng(1)
T = tris.reshape(-1, 12)
def make_rays(n):
    out = []
    for _ in range(n):
        t = T[rng.integers(0, T.shape[0])]
        v0, e1, e2 = t[0:3].astype(np.float64), t[4:7].astype(np.float64), t[8:11].astype(np.float64)
        p = v0 + (e1 + e2) / 3
        nrm = np.cross(e1, e2); nrm /= np.linalg.norm(nrm)
        u1, u2 = rng.random(), rng.random()
        r = np.sqrt(u2); phi = 2 * np.pi * u1
        up = np.array([0, 1.0, 0]) if abs(nrm[1]) < 0.99 else np.array([1.0, 0, 0])
        tx = np.cross(up, nrm); tx /= np.linalg.norm(tx); bx = np.cross(nrm, tx)
        d = r * np.cos(phi) * tx + r * np.sin(phi) * bx + np.sqrt(max(0, 1 - u2)) * nrm
        d /= np.linalg.norm(d)
        out.append((p + nrm * 0.002, d, 0.8))
    return out
rays = make_rays(int(sys.argv[2]) if len(sys.argv) > 2 else 600)
for name, w4 in (("median 4-wide", collapse_median()), ("SAH 4-wide", build_sah(4)), ("SAH 8-wide", build_sah(8))):
    t0 = time.time()
    r = np.array([traverse(w4, *ray) for ray in rays], dtype=np.float64).mean(0)
    print(f"{name:16s} nodes {len(w4):7d} depth {depth_of(w4):3d}  per AO-like ray (unoccluded-style full traversal): inner visits {r[0]:.1f} leaf visits {r[1]:.1f} tri tests {r[2]:.1f}   ({time.time()-t0:.0f}s)")

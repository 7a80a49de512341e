"""The arithmetic identity the fused closest-hit records rest on (csrc/rt_wave.hip `slab_union`, csrc/rt_api.hip rt_upload_bvh "fused closest-hit records").

The reference's step at a node N tests the boxes of N's children A and B (rt_bvh.glsl:226-239, aabbHit :124-134).  The fused record of N holds the boxes of A's
and B's CHILDREN only; A's box is the union of its children's (checked at upload), and the kernel derives A's slab values from the children's:

    per axis   tsm(A) = min(tsm(A1), tsm(A2))      tbg(A) = max(tbg(A1), tbg(A2))

with v_min_f32 / v_max_f32 semantics (a NaN operand is dropped).  This file pins that against the slab test evaluated on the union box itself -- float32,
signed zeros, +-inf reciprocal directions, the NaN of 0 * inf (an origin coordinate exactly on a box plane of an axis the ray does not move along) and the
NaN box of an absent child included.  CPU only (numpy float32 = IEEE binary32 RNE, np.fmin / np.fmax = NaN-dropping min / max)."""
import itertools

import numpy as np

f32 = np.float32


def parts(ro, rdinv, bmin, bmax):
    with np.errstate(invalid="ignore", over="ignore"):
        t0 = ((bmin - ro).astype(f32) * rdinv).astype(f32)
        t1 = ((bmax - ro).astype(f32) * rdinv).astype(f32)
    return np.fmin(t0, t1), np.fmax(t0, t1)


def evaluate(sm, bg):
    """slab(): tmin = max(max(sx, sy), max(sz, 0)), tmax = min(min(bx, by), bz), hit = tmax >= tmin"""
    tmin = np.fmax(np.fmax(sm[..., 0], sm[..., 1]), np.fmax(sm[..., 2], f32(0.0)))
    tmax = np.fmin(np.fmin(bg[..., 0], bg[..., 1]), bg[..., 2])
    with np.errstate(invalid="ignore"):
        return tmax >= tmin, tmin


def same(a, b):
    """equal as floats (so -0 == +0: only comparisons ever look at these values), or both NaN"""
    return np.all((a == b) | (np.isnan(a) & np.isnan(b)))


def test_union_parts_per_axis_exhaustive():
    nan = f32(np.nan)
    coords = np.array([-2.0, -1.0, -0.0, 0.0, 1e-30, 0.5, 1.0, 2.0, 3.0e38, -3.0e38], f32)
    ros = np.array([0.0, -0.0, 1.0, -1.0, 0.5, 2.0, 3.0e38], f32)
    rdinvs = np.array([np.inf, -np.inf, 0.5, -3.0, 1.0, 3.0e38, -3.0e38, 1e-38, -1e-38], f32)
    boxes = [(a, b) for a, b in itertools.product(coords, coords) if a <= b] + [(nan, nan)]   # + an absent child's NaN box
    b1 = np.array(boxes, f32)
    n = len(boxes)
    lo1 = np.repeat(b1[:, 0], n); hi1 = np.repeat(b1[:, 1], n)
    lo2 = np.tile(b1[:, 0], n); hi2 = np.tile(b1[:, 1], n)
    ulo, uhi = np.fmin(lo1, lo2), np.fmax(hi1, hi2)       # the union box as rt_upload_bvh checks it: min / max of the children's coordinates
    checked = 0
    for ro in ros:
        for rdinv in rdinvs:
            s1, g1 = parts(ro, rdinv, lo1, hi1)
            s2, g2 = parts(ro, rdinv, lo2, hi2)
            su, gu = parts(ro, rdinv, ulo, uhi)
            assert same(np.fmin(s1, s2), su), (ro, rdinv)
            assert same(np.fmax(g1, g2), gu), (ro, rdinv)
            checked += lo1.size
    assert checked > 200000


def test_union_slab_decisions_random_boxes_and_rays():
    rng = np.random.default_rng(7)
    n = 400000
    c1 = rng.uniform(-2, 2, (n, 3)).astype(f32); e1 = rng.uniform(0, 1, (n, 3)).astype(f32)
    c2 = (c1 + rng.uniform(-1, 1, (n, 3))).astype(f32); e2 = rng.uniform(0, 1, (n, 3)).astype(f32)
    lo1, hi1, lo2, hi2 = c1 - e1, c1 + e1, c2 - e2, c2 + e2
    flat = rng.random((n, 3)) < 0.05
    hi1 = np.where(flat, lo1, hi1).astype(f32)             # flat boxes (axis-aligned triangles)
    ro = rng.uniform(-4, 4, (n, 3)).astype(f32)
    snap = rng.random((n, 3)) < 0.2                         # origins exactly on box planes
    pick = rng.integers(0, 4, (n, 3))
    planes = np.stack([lo1, hi1, lo2, hi2], axis=-1)
    ro = np.where(snap, np.take_along_axis(planes, pick[..., None], axis=-1)[..., 0], ro).astype(f32)
    rd = rng.normal(size=(n, 3)).astype(f32)
    zero = rng.random((n, 3)) < 0.2
    rd = np.where(zero, np.where(rng.random((n, 3)) < 0.5, f32(0.0), f32(-0.0)), rd).astype(f32)
    with np.errstate(divide="ignore"):
        rdinv = (f32(1.0) / rd).astype(f32)
    # absent second child in a tenth of the cases
    absent = rng.random(n) < 0.1
    lo2 = np.where(absent[:, None], f32(np.nan), lo2).astype(f32); hi2 = np.where(absent[:, None], f32(np.nan), hi2).astype(f32)
    s1, g1 = parts(ro, rdinv, lo1, hi1)
    s2, g2 = parts(ro, rdinv, lo2, hi2)
    su, gu = parts(ro, rdinv, np.fmin(lo1, lo2), np.fmax(hi1, hi2))
    hit_d, tmin_d = evaluate(np.fmin(s1, s2), np.fmax(g1, g2))
    hit_u, tmin_u = evaluate(su, gu)
    assert np.array_equal(hit_d, hit_u)
    assert same(tmin_d, tmin_u)
    assert hit_u.sum() > n // 100 and (~hit_u).sum() > n // 100
    # Monotonicity -- a child that passes its own test implies the union passes, with tmin(union) <= tmin(child) -- holds wherever no slab value of the
    # child is NaN.  It does NOT hold in the one corner left: a FLAT child box with the ray exactly in its plane along an axis it does not move along gives
    # NaN on both planes of that axis, the axis drops out and the child passes, while the (thicker) parent sees the origin ON its boundary, (-inf, -inf) or
    # (+inf, +inf), and fails.  Which is why k_trace<FUSE> evaluates the parent's test (slab_eval of the union) as the reference does instead of inferring
    # it from the children.  (For the any-hit walk of the four-wide nodes the corner is harmless: every triangle of such a box lies in the ray's plane,
    # det == 0, rt_bvh.glsl:158 rejects it.)
    hit_c, tmin_c = evaluate(s1, g1)
    clean = ~(np.isnan(s1).any(axis=-1) | np.isnan(g1).any(axis=-1))
    assert np.all(hit_u[hit_c & clean])
    assert np.all(tmin_u[hit_c & clean] <= tmin_c[hit_c & clean])
    assert (hit_c & ~hit_u).sum() > 0 and not np.any(hit_c & ~hit_u & clean)     # the corner exists, and only there


def test_builder_boxes_are_unions_of_their_childrens():
    """what rt_upload_bvh checks before it builds the fused records, on the host builder's output (the oracle's build_bvh = bvh.cpp:41-137)"""
    import scenes
    nodes, tris = scenes.bunny_bvh(3)
    left = (nodes[:, 3] + 0.5).astype(int); right = (nodes[:, 7] + 0.5).astype(int); count = (nodes[:, 9] + 0.5).astype(int)
    inner = np.nonzero(count == 0)[0]
    assert inner.size > 100
    lo = np.minimum(nodes[left[inner], 0:3], nodes[right[inner], 0:3])
    hi = np.maximum(nodes[left[inner], 4:7], nodes[right[inner], 4:7])
    assert np.array_equal(lo, nodes[inner, 0:3]) and np.array_equal(hi, nodes[inner, 4:7])


def test_implicit_records_preorder_address():
    """The address arithmetic of the implicit records (csrc/rt_api.hip "implicit records", csrc/rt_wave.hip k_trace<.., IMPL>): in a perfect binary tree whose leaves sit at
    depth D, the inner node reached by the left / right turns p (as a binary number, d bits) at depth d has the pre-order position  d - popcount(p) + (p << (D - d))  among
    the inner nodes -- a left child sits next to its parent, a right child behind the whole left subtree -- and the positions of all inner nodes are a permutation of
    0 .. 2^D - 2.  Checked against an explicit pre-order walk."""
    for D in range(1, 9):
        want = {}
        counter = [0]

        def walk(d, p):
            if d == D:
                return
            want[(d, p)] = counter[0]
            counter[0] += 1
            walk(d + 1, 2 * p)
            walk(d + 1, 2 * p + 1)

        walk(0, 0)
        assert counter[0] == (1 << D) - 1
        got = {(d, p): d - bin(p).count("1") + (p << (D - d)) for (d, p) in want}
        assert got == want, D
        assert sorted(got.values()) == list(range((1 << D) - 1))
        # four-wide form: the even-depth nodes keep their binary pre-order position (odd-depth slots of the array stay empty), children (d + 2, 4p + j)
        for (d, p), at in want.items():
            if d % 2 == 0 and d + 2 < D:
                for j in range(4):
                    assert (d + 2, 4 * p + j) in want

"""Procedural triangle meshes standing in for the reference's `models/*.obj`.

The reference git-ignores `*.obj` (its .gitignore:20), so `models/bunny_lp.obj` named at
src/app/application.cpp:260-265 does not exist anywhere; SURVEY.md 8d fixes the stand-ins:
  * "bunny": icosphere, 6 subdivisions (81 920 triangles), radially displaced by 3 octaves of
    lattice value noise, seed 0x5EED, raw radius ~1 (0.5 after `defaultBvhTransform`);
  * "1M": 16 noise-displaced UV spheres of 62 500 triangles on a 4x4 grid.
Everything is float32 numpy with integer hashing only (no libm), so the meshes are bit-identical
on every machine.  Meshes can be written as .obj and read back by rt_load_obj.
"""
from __future__ import annotations

import numpy as np

BUNNY_SEED = 0x5EED


def icosphere(subdiv: int):
    """Unit icosphere: (verts [N,3] float32, indices [M*3] uint32), 20 * 4**subdiv triangles."""
    t = (1.0 + 5.0 ** 0.5) / 2.0
    v = np.array([[-1, t, 0], [1, t, 0], [-1, -t, 0], [1, -t, 0], [0, -1, t], [0, 1, t], [0, -1, -t], [0, 1, -t],
                  [t, 0, -1], [t, 0, 1], [-t, 0, -1], [-t, 0, 1]], np.float64)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    f = np.array([[0, 11, 5], [0, 5, 1], [0, 1, 7], [0, 7, 10], [0, 10, 11], [1, 5, 9], [5, 11, 4], [11, 10, 2], [10, 7, 6],
                  [7, 1, 8], [3, 9, 4], [3, 4, 2], [3, 2, 6], [3, 6, 8], [3, 8, 9], [4, 9, 5], [2, 4, 11], [6, 2, 10],
                  [8, 6, 7], [9, 8, 1]], np.int64)
    for _ in range(subdiv):
        n = v.shape[0]
        e = np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]], axis=0)
        e.sort(axis=1)
        key = e[:, 0] * n + e[:, 1]
        uniq, inv = np.unique(key, return_inverse=True)
        a, b = uniq // n, uniq % n
        mid = v[a] + v[b]
        mid /= np.linalg.norm(mid, axis=1, keepdims=True)
        v = np.concatenate([v, mid], axis=0)
        m = n + inv.reshape(3, -1).T          # midpoints of (01, 12, 20) per face
        f = np.concatenate([np.stack([f[:, 0], m[:, 0], m[:, 2]], 1), np.stack([f[:, 1], m[:, 1], m[:, 0]], 1),
                            np.stack([f[:, 2], m[:, 2], m[:, 1]], 1), np.stack([m[:, 0], m[:, 1], m[:, 2]], 1)], axis=0)
    return v.astype(np.float32), f.astype(np.uint32).reshape(-1)


def _hash3(ix, iy, iz, seed):
    h = (ix.astype(np.uint32) * np.uint32(0x9E3779B1)) ^ (iy.astype(np.uint32) * np.uint32(0x85EBCA77)) ^ \
        (iz.astype(np.uint32) * np.uint32(0xC2B2AE3D)) ^ np.uint32(seed)
    h ^= h >> np.uint32(15)
    h *= np.uint32(0x2C1B3C6D)
    h ^= h >> np.uint32(12)
    h *= np.uint32(0x297A2D39)
    h ^= h >> np.uint32(15)
    return h


def value_noise(p, seed):
    """Trilinear lattice value noise in [0,1), float32, p [N,3]."""
    p = p.astype(np.float32)
    fl = np.floor(p)
    fr = p - fl
    w = fr * fr * (np.float32(3.0) - np.float32(2.0) * fr)
    i = fl.astype(np.int64)
    acc = np.zeros(p.shape[0], np.float32)
    for dz in (0, 1):
        for dy in (0, 1):
            for dx in (0, 1):
                h = _hash3(i[:, 0] + dx, i[:, 1] + dy, i[:, 2] + dz, seed)
                val = (h >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)
                wx = w[:, 0] if dx else np.float32(1.0) - w[:, 0]
                wy = w[:, 1] if dy else np.float32(1.0) - w[:, 1]
                wz = w[:, 2] if dz else np.float32(1.0) - w[:, 2]
                acc += val * wx * wy * wz
    return acc


def displace(verts, seed, octaves=3, amplitude=0.25, base_freq=2.0):
    """Radial displacement r = 1 - amplitude/2 + amplitude * fbm(p) of unit-sphere points."""
    v = verts.astype(np.float32)
    n = np.zeros(v.shape[0], np.float32)
    amp, freq, norm = np.float32(1.0), np.float32(base_freq), np.float32(0.0)
    for o in range(octaves):
        n += amp * value_noise(v * freq + np.float32(17.0 * (o + 1)), (seed + 0x9E37 * o) & 0xFFFFFFFF)
        norm += amp
        amp *= np.float32(0.5)
        freq *= np.float32(2.0)
    n /= norm
    r = np.float32(1.0 - 0.5 * amplitude) + np.float32(amplitude) * n
    return (v * r[:, None]).astype(np.float32)


def bunny_standin(subdiv=6, seed=BUNNY_SEED):
    """(verts, indices): 20*4**subdiv triangles; subdiv=6 is BASELINE config 2's mesh."""
    v, f = icosphere(subdiv)
    return displace(v, seed), f


def uv_sphere(nu, nv):
    """UV sphere with nu x nv quads -> 2*nu*nv triangles (degenerate pole triangles kept, like a naive exporter)."""
    u = np.arange(nu + 1, dtype=np.float64) / nu * 2.0 * np.pi
    w = np.arange(nv + 1, dtype=np.float64) / nv * np.pi
    uu, ww = np.meshgrid(u, w)
    v = np.stack([np.sin(ww) * np.cos(uu), np.cos(ww), np.sin(ww) * np.sin(uu)], -1).reshape(-1, 3)
    j, i = np.meshgrid(np.arange(nv), np.arange(nu), indexing="ij")
    a = (j * (nu + 1) + i).reshape(-1)
    b, c, d = a + 1, a + (nu + 1), a + (nu + 2)
    f = np.concatenate([np.stack([a, c, b], 1), np.stack([b, c, d], 1)], 0)
    return np.round(v, 9).astype(np.float32), f.astype(np.uint32).reshape(-1)


def million_triangle_scene(n_objects=16, nu=250, nv=125):
    """n_objects noise-displaced UV spheres (2*nu*nv tris each) on a 4x4 grid,
    x in [-6,6], z in [-12,0], radius 1, centres at y=1.2.  Returns world-space (verts, indices)."""
    verts, idx, base = [], [], 0
    side = int(round(n_objects ** 0.5))
    for k in range(n_objects):
        v, f = uv_sphere(nu, nv)
        v = displace(v, k + 1, amplitude=0.3)
        gx, gz = k % side, k // side
        cx = -6.0 + 12.0 * (gx + 0.5) / side
        cz = -12.0 + 12.0 * (gz + 0.5) / side
        v = (v * np.float32(1.2) + np.array([cx, 1.4, cz], np.float32)).astype(np.float32)
        verts.append(v)
        idx.append(f + np.uint32(base))
        base += v.shape[0]
    return np.concatenate(verts, 0), np.concatenate(idx, 0)


def write_obj(path, verts, indices):
    f = np.asarray(indices).reshape(-1, 3) + 1
    with open(path, "w") as fh:
        fh.write("# generated by opengl-raytracing_amd.meshgen\n")
        for v in np.asarray(verts, np.float32):
            fh.write("v %.9g %.9g %.9g\n" % (v[0], v[1], v[2]))
        for t in f:
            fh.write("f %d %d %d\n" % (t[0], t[1], t[2]))

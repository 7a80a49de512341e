"""Generates tests/golden/*.npz with the parity oracle (oracle/liborc.so).

The reference ships no golden images or test vectors and its GL renderer cannot run headless here
(SURVEY.md section 4 / 8c), so these fixtures pin the ORACLE against regressions ("parity unpinned" by the
reference; the KATs of SURVEY.md 8c are checked separately in tests/test_oracle_kat.py).  Each
fixture stores its complete inputs (uniform block bytes, BVH arrays, cube-map faces) so that the
check does not depend on any generator being bit-stable across machines.

    python tests/golden/make_golden.py        # rewrites the fixtures
"""
import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
for p in (str(ROOT), str(ROOT / "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import opengl_raytracing_amd as rt  # noqa: E402
import oracle as orc  # noqa: E402


def ubytes(u):
    return np.frombuffer(bytes(u), np.uint8).copy()


def tiny_env(n, seed):
    return np.random.default_rng(seed).integers(0, 256, size=(6, n, n, 3), dtype=np.uint8)


def run_frames(params, cam, w, h, use_bvh, nodes, tris, faces, frames):
    prev, us, outs, cnts = None, [], [], []
    for f in range(frames):
        u = orc.frame_uniforms(params, cam, w, h, f, use_bvh, 0 if nodes is None else nodes.shape[0], 0 if tris is None else tris.shape[0])
        o, c = orc.render(u, nodes, tris, faces, prev)
        us.append(ubytes(u)); outs.append(o); cnts.append(np.array(c.as_tuple(), np.uint64))
        prev = o[0]
    return us, outs, cnts


def save(name, us, outs, cnts, nodes, tris, faces):
    d = {"uniforms": np.stack(us), "counters": np.stack(cnts)}
    for f, o in enumerate(outs):
        for k, a in zip(("color", "motion", "gpos", "gnrm"), o):
            d[f"{k}{f}"] = a
    if nodes is not None:
        d["nodes12"], d["tris12"] = nodes, tris
    if faces is not None:
        d["env"] = faces
    np.savez_compressed(HERE / f"{name}.npz", **d)
    print(name, {k: v.shape for k, v in d.items() if k in ("uniforms", "color0", "nodes12", "env")})


def main():
    # 1. BASELINE config 1 in miniature: analytic scene, gradient sky, 64x64, frames 0..2 (TAA history)
    p = orc.default_render_params(); p.enableEnvMap = 0
    cam = orc.default_camera(); cam.aspect = 1.0
    us, outs, cnts = run_frames(p, cam, 64, 64, False, None, None, None, 3)
    save("analytic_gradient_64", us, outs, cnts, None, None, None)
    # 2. analytic scene with a cube map, 2 spp
    p = orc.default_render_params(); p.sppPerFrame = 2
    faces = tiny_env(8, 11)
    us, outs, cnts = run_frames(p, cam, 48, 48, False, None, None, faces, 2)
    save("analytic_env_48", us, outs, cnts, None, None, faces)
    # 3. BVH scene: icosphere subdiv 2 (320 triangles) displaced, close-up, 2 spp, GI + AO, frames 0..1
    v, f = rt.meshgen.bunny_standin(2)
    tris9 = orc.gather_triangles(v, f)
    nodes, tris = orc.build_bvh(tris9)
    p = orc.default_render_params(); p.sppPerFrame = 2
    cam = orc.default_camera(); cam.pos[0], cam.pos[1], cam.pos[2] = -2.0, 1.5, 1.0; cam.yaw, cam.pitch, cam.aspect = -90.0, 0.0, 1.5
    us, outs, cnts = run_frames(p, cam, 48, 32, True, nodes, tris, faces, 2)
    d_extra = {"tris9": tris9}
    save("bvh_closeup_48x32", us, outs, cnts, nodes, tris, faces)
    np.savez_compressed(HERE / "bvh_build_320.npz", tris9=tris9, nodes12=nodes, tris12=tris)


if __name__ == "__main__":
    main()

"""Container-only: compare the oracle with the reference's own GLSL (SwiftShader, oracle/glsl_ref.py) on many random
analytic-scene configurations -- cameras, materials, lights, toggles, spp, sizes, static and moving -- and write the
summary tests/golden/glsl_sweep_summary.txt.  (The committed fixtures glsl_*.npz are five hand-picked cases; this sweep is
the evidence that they are typical.)    python tests/golden/glsl_sweep.py [cases]"""
import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
for p in (str(ROOT), str(ROOT / "tests"), str(ROOT / "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

import opengl_raytracing_amd as rt  # noqa: E402
import oracle as orc  # noqa: E402
from glsl_ref import GlslReference  # noqa: E402
from make_glsl_golden import ring_env, h2f  # noqa: E402


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    g = GlslReference()
    worst = {k: (1.0, 0.0, -1) for k in ("color", "motion", "gpos", "gnrm")}   # (min exact, max rmse, seed)
    tot = {k: [0, 0] for k in worst}
    rmses = {k: [] for k in worst}
    flips = 0   # pixels where one side hit and the other missed (gpos.w differs)
    npx = 0
    lines = []
    for seed in range(cases):
        rng = np.random.default_rng(5000 + seed)
        W, H = int(rng.integers(17, 80)), int(rng.integers(17, 60))
        p = orc.default_render_params()
        p.sppPerFrame = int(rng.choice([1, 2, 3]))
        for name in ("enableGI", "enableAO", "enableTAA", "enableJitter", "sunEnabled", "skyEnabled", "pointLightEnabled", "enableEnvMap",
                     "matGlassEnabled", "matMirrorEnabled"):
            setattr(p, name, int(rng.random() < 0.75))
        p.matGlassIOR = float(rng.uniform(1.0, 2.2)); p.matGlassDistortion = float(rng.uniform(0.0, 0.5))
        p.matMirrorGloss = float(rng.uniform(0.0, 1.0)); p.matAlbedoGloss = float(rng.uniform(1.0, 128.0))
        p.matAlbedoSpecStrength = float(rng.uniform(0.0, 1.0))
        p.pointLightPos[0], p.pointLightPos[1], p.pointLightPos[2] = float(rng.normal(0, 2)), float(rng.uniform(0.2, 4)), float(rng.normal(0, 2))
        p.aoSamples = int(rng.integers(1, 6))
        faces = ring_env(int(rng.choice([3, 8])), int(rng.integers(100))) if p.enableEnvMap else None
        cams = []
        for k in range(3):
            c = orc.default_camera(); c.aspect = W / H
            c.pos[0] += float(rng.normal(0, 1.5)); c.pos[1] = float(rng.uniform(0.1, 5.0)); c.pos[2] += float(rng.normal(0, 1.5))
            c.yaw += float(rng.normal(0, 40)); c.pitch += float(rng.normal(0, 25)); c.fov = float(rng.uniform(25, 110))
            cams.append(c)
        moving = bool(rng.random() < 0.5)
        prev, prev_vp = None, None
        for frame in range(3):
            cam = cams[frame] if moving else cams[0]
            u = orc.frame_uniforms(p, cam, W, H, frame, False, prev_vp=prev_vp, env_loaded=faces is not None)
            prev_vp = orc.mat4_mul(orc.camera_proj(cam), orc.camera_view(cam))
            got = g.render(u, None, None, faces, prev)
            want, _ = orc.render(u, None, None, faces, prev)
            for k, a, b in zip(("color", "motion", "gpos", "gnrm"), got, want):
                fa, fb = h2f(a), h2f(b)
                ok = np.isfinite(fa) & np.isfinite(fb)
                d = np.where(ok, (fa - fb) / np.maximum(1.0, np.abs(fb)), 0.0)   # relative above 1: reprojection with w ~ 0 gives huge NDC values
                rmse = float(np.sqrt(np.mean(d * d)))
                exact = float(np.mean(a == b))
                tot[k][0] += int(np.sum(a == b)); tot[k][1] += a.size
                rmses[k].append(rmse)
                if k == 'gpos':
                    flips += int(np.sum(fa[:, :, 3] != fb[:, :, 3])); npx += fa.shape[0] * fa.shape[1]
                if exact < worst[k][0] or rmse > worst[k][1]:
                    worst[k] = (min(exact, worst[k][0]), max(rmse, worst[k][1]), seed)
            prev = got[0]
    lines.append(f"{cases} random analytic-scene configurations x 3 frames, reference GLSL ({g.version}) vs oracle")
    for k in worst:
        lines.append(f"  {k:6s}: {tot[k][0] / tot[k][1] * 100:.3f} % of all values bit-identical; worst frame: {worst[k][0] * 100:.2f} % identical, RMSE {worst[k][1]:.3e} (last at seed {worst[k][2]})")
    for k in worst:
        r = np.array(rmses[k])
        lines.append(f"  {k:6s} RMSE per frame: median {np.median(r):.2e}, 90th pct {np.percentile(r, 90):.2e}, frames above 1e-4: {int(np.sum(r > 1e-4))} of {r.size}")
    lines.append(f"  (RMSE of (a - b) / max(1, |b|).)  The primary hit/miss decision differs on {flips} of {npx} pixels; frames above 1e-4 are "
                 "frames with isolated pixels at secondary discontinuities (glass / mirror paths, shadow edges) where the last bit decides")
    (HERE / "glsl_sweep_summary.txt").write_text("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()

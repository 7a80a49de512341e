"""Headless C++ host (opengl-raytracing_amd/rt_cli, built from csrc/rt_cli.cpp): the reference's start-up + frame loop
driven purely through the C ABI.  CPU: PNG writer round trip.  GPU: the CLI's PNG equals the frames the Python harness
renders through the same library (and hence the oracle's, by the other parity tests)."""
import glob
import os
import subprocess

import numpy as np
import pytest

import opengl_raytracing_amd as rt
import scenes

CLI = scenes.ROOT / "opengl-raytracing_amd" / "rt_cli"


def test_png_writer_round_trip(tmp_path):
    rng = np.random.default_rng(0)
    for shape in ((7, 5, 4), (6, 9, 3), (4, 4)):
        a = rng.integers(0, 256, size=shape, dtype=np.uint8)
        rt.save_png(tmp_path / "a.png", a)
        assert np.array_equal(rt.load_png(tmp_path / "a.png").reshape(a.shape), a)
        rt.save_png(tmp_path / "f.png", a, flip_y=True)
        assert np.array_equal(rt.load_png(tmp_path / "f.png").reshape(a.shape), a[::-1])
    from PIL import Image
    a = rng.integers(0, 256, size=(5, 6, 3), dtype=np.uint8)
    rt.save_png(tmp_path / "p.png", a)
    assert np.array_equal(np.asarray(Image.open(tmp_path / "p.png")), a)
    with pytest.raises(rt.RtError):
        rt.save_png(tmp_path / "nodir" / "x.png", a)


def test_cli_is_built_and_prints_usage():
    assert CLI.exists(), "run __graft_entry__.build()"
    out = subprocess.run([str(CLI), "--help"], capture_output=True, text=True)
    assert out.returncode == 0 and "usage: rt_cli" in out.stderr


def test_cli_ranks_launcher_dry_run(tmp_path):
    """rt_cli --ranks N --dry-run (no GPU call anywhere): N children forked, rank r bound to --devices[r], the 128-byte id written by
    rank 0 (atomically, into a file named after the launcher's pid) and read back intact by every other rank, the file removed."""
    out = subprocess.run([str(CLI), "--ranks", "3", "--devices", "5,2,7", "--dry-run", "--out", str(tmp_path / "d")], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = sorted(ln for ln in out.stdout.splitlines() if ln.startswith("[DRY]"))
    assert lines == ["[DRY] rank 0 of 3 device 5 id ok", "[DRY] rank 1 of 3 device 2 id ok", "[DRY] rank 2 of 3 device 7 id ok"]
    assert not glob.glob(str(tmp_path / "d.rccl_id*"))
    bad = subprocess.run([str(CLI), "--ranks", "2", "--devices", "0", "--dry-run"], capture_output=True, text=True, timeout=60)
    assert bad.returncode == 2 and "--devices needs 2 entries" in bad.stderr


def test_cli_ranks_launcher_stops_the_survivors_when_a_rank_dies(tmp_path):
    """One rank exits non-zero while another sits in a (simulated) collective forever: the launcher ends the survivor and returns
    non-zero instead of blocking in waitpid on the hung rank."""
    env = dict(os.environ, RT_CLI_DRY_FAIL_RANK="1", RT_CLI_DRY_HANG_RANK="2")
    out = subprocess.run([str(CLI), "--ranks", "3", "--dry-run", "--out", str(tmp_path / "d")], capture_output=True, text=True, timeout=60, env=env)
    assert out.returncode == 1 and "ranks failed or were stopped" in out.stderr, out.stdout + out.stderr
    assert not glob.glob(str(tmp_path / "d.rccl_id*"))


@pytest.mark.gpu
def test_cli_renders_the_same_frames_as_the_python_harness(tmp_path):
    W, H, frames, spp = 160, 96, 3, 2
    v, f = rt.meshgen.bunny_standin(3)
    obj = tmp_path / "blob.obj"
    rt.meshgen.write_obj(obj, v, f)
    env = scenes.ASSETS / "Sky_16.png"
    cam = "-2,1.5,1.0,-90,0"
    for mode in ("bvh", "analytic"):
        args = [str(CLI), "--env", str(env), "--size", f"{W}x{H}", "--spp", str(spp), "--frames", str(frames), "--out", str(tmp_path / mode)]
        args += ["--obj", str(obj), "--cam", cam] if mode == "bvh" else ["--analytic"]
        out = subprocess.run(args, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stdout + out.stderr
        got = rt.load_png(tmp_path / f"{mode}.png")
        # the same thing through the Python harness
        p = rt.default_render_params()
        p.sppPerFrame = spp
        c = scenes.camera("closeup" if mode == "bvh" else "default", aspect=W / H)
        with rt.Renderer() as r:
            if mode == "bvh":
                v2, f2 = rt.load_obj(obj)
                nodes, tris = rt.build_bvh(rt.gather_triangles(v2, f2))
                r.upload_bvh(nodes, tris)
            r.upload_env(scenes.env_faces("Sky_16"))
            r.resize(W, H)
            for _ in range(frames):
                r.render_ray(p, c, use_bvh=(mode == "bvh"))
            want = r.present(p)[::-1]          # PNG rows are top-down
        assert np.array_equal(got, want), mode


@pytest.mark.gpu
def test_cli_merges_several_obj_files_and_dumps_the_targets(tmp_path):
    """several .obj files -> one triangle soup -> build_bvh; --dump-targets writes the four targets as PFM."""
    W, H = 96, 64
    va, fa = rt.meshgen.bunny_standin(2)
    vb, fb = rt.meshgen.bunny_standin(1)
    vb = vb * 0.5 + np.array([0.6, 0.3, 0.0], np.float32)
    rt.meshgen.write_obj(tmp_path / "a.obj", va, fa)
    rt.meshgen.write_obj(tmp_path / "b.obj", vb, fb)
    cam = "-2,1.5,1.0,-90,0"
    args = [str(CLI), "--obj", str(tmp_path / "a.obj"), "--obj", str(tmp_path / "b.obj"), "--no-env", "--size", f"{W}x{H}", "--spp", "2",
            "--frames", "2", "--cam", cam, "--dump-targets", "--out", str(tmp_path / "two")]
    out = subprocess.run(args, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert f"[BVH] {(fa.size + fb.size) // 3} triangles" in out.stdout, out.stdout

    def read_pfm(path):
        with open(path, "rb") as f:
            assert f.readline().strip() == b"PF"
            w, h = map(int, f.readline().split())
            assert float(f.readline()) < 0                      # little endian
            return np.frombuffer(f.read(), "<f4").reshape(h, w, 3)

    # the same scene through the Python harness
    p = rt.default_render_params()
    p.sppPerFrame = 2
    p.enableEnvMap = 0
    c = scenes.camera("closeup", aspect=W / H)
    tris9 = np.concatenate([rt.gather_triangles(*rt.load_obj(tmp_path / "a.obj")), rt.gather_triangles(*rt.load_obj(tmp_path / "b.obj"))])
    nodes, tris = rt.build_bvh(tris9)
    with rt.Renderer() as r:
        r.upload_bvh(nodes, tris)
        r.resize(W, H)
        for _ in range(2):
            r.render_ray(p, c, use_bvh=True)
        for which, name, ch in ((rt.RT_TARGET_COLOR, "color", 3), (rt.RT_TARGET_MOTION, "motion", 2), (rt.RT_TARGET_GPOS, "gpos", 3), (rt.RT_TARGET_GNRM, "gnrm", 3)):
            want = r.read_target(which, rt.RT_FORMAT_F32)
            got = read_pfm(tmp_path / f"two_{name}.pfm")
            assert np.array_equal(got[:, :, :ch], want[:, :, :ch]), name


def test_cli_rejects_bad_scene_files(tmp_path):
    bad = tmp_path / "bad.json"
    bad.write_text('{"sppPerFrame": 2, "noSuchField": 1}')
    out = subprocess.run([str(CLI), "--scene", str(bad)], capture_output=True, text=True, timeout=60)
    assert out.returncode == 2 and "noSuchField" in out.stderr
    bad.write_text('{"sppPerFrame": ')
    out = subprocess.run([str(CLI), "--scene", str(bad)], capture_output=True, text=True, timeout=60)
    assert out.returncode == 2


@pytest.mark.gpu
def test_cli_scene_file_equals_the_flags(tmp_path):
    """a JSON scene file (RenderParams field names + camera / size / frames / obj / env) drives the same frames as flags."""
    import json
    v, f = rt.meshgen.bunny_standin(2)
    obj = tmp_path / "blob.obj"
    rt.meshgen.write_obj(obj, v, f)
    scene = {"obj": [str(obj)], "env": str(scenes.ASSETS / "Sky_16.png"), "size": [120, 72], "frames": 2, "sppPerFrame": 3, "enableAO": False,
             "sunEnabled": 1, "sunYaw": 30.0, "sunColor": [1.0, 0.9, 0.8], "exposure": 1.25, "out": str(tmp_path / "a"),
             "camera": {"pos": [-2, 1.5, 1.0], "yaw": -90, "pitch": 0, "fov": 55.0}}
    (tmp_path / "scene.json").write_text(json.dumps(scene, indent=1))
    a = subprocess.run([str(CLI), "--scene", str(tmp_path / "scene.json")], capture_output=True, text=True, timeout=300)
    assert a.returncode == 0, a.stdout + a.stderr
    p = rt.default_render_params()
    p.sppPerFrame, p.enableAO, p.sunEnabled, p.sunYaw, p.exposure = 3, 0, 1, 30.0, 1.25
    p.sunColor[0], p.sunColor[1], p.sunColor[2] = 1.0, 0.9, 0.8
    c = scenes.camera("closeup", aspect=120 / 72)
    c.fov = 55.0
    with rt.Renderer() as r:
        r.upload_bvh(*rt.build_bvh(rt.gather_triangles(*rt.load_obj(obj))))
        r.upload_env(scenes.env_faces("Sky_16"))
        r.resize(120, 72)
        for _ in range(2):
            r.render_ray(p, c, use_bvh=True)
        want = r.present(p)[::-1]
    assert np.array_equal(rt.load_png(tmp_path / "a.png"), want)


@pytest.mark.gpu
def test_cli_point_light_orbit_follows_the_main_loop(tmp_path):
    """application.cpp:341-348, 538-553: the orbit advances the light's yaw by speed * dt per frame and, being dynamic geometry,
    resets the accumulation after every frame."""
    W, H, frames, dt = 96, 64, 3, 0.05
    args = [str(CLI), "--analytic", "--no-env", "--size", f"{W}x{H}", "--frames", str(frames), "--dt", str(dt), "--scene", str(tmp_path / "s.json"), "--out", str(tmp_path / "o")]
    (tmp_path / "s.json").write_text('{"pointLightOrbitEnabled": 1, "pointLightOrbitSpeed": 90.0, "pointLightOrbitRadius": 1.5, "pointLightEnabled": 1}')
    out = subprocess.run(args, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    p = rt.default_render_params()
    p.enableEnvMap, p.pointLightOrbitEnabled, p.pointLightOrbitSpeed, p.pointLightOrbitRadius, p.pointLightEnabled = 0, 1, 90.0, 1.5, 1
    c = scenes.camera("default", aspect=W / H)
    with rt.Renderer() as r:
        r.resize(W, H)
        for f in range(frames):
            p.pointLightYaw = np.float32(np.float32(p.pointLightYaw) + np.float32(90.0) * np.float32(dt))
            r.render_ray(p, c, use_bvh=False)
            if f + 1 < frames:
                r.reset_accum()
        want = r.present(p)[::-1]
    assert np.array_equal(rt.load_png(tmp_path / "o.png"), want)


@pytest.mark.gpu
def test_cli_ranks_mode_equals_the_single_process_run(tmp_path):
    """rt_cli --ranks 1: a forked child per GPU, RCCL id handed over through a file, rt_comm_init, rt_gather_frame every k-th
    frame, present from the gathered targets, PFM dumps through rt_read_gathered -- byte-identical to the plain run.  (More
    than one rank needs more than one GPU: RCCL refuses two ranks on one device.)"""
    W, H, frames = 128, 80, 5
    v, f = rt.meshgen.bunny_standin(3)
    obj = tmp_path / "blob.obj"
    rt.meshgen.write_obj(obj, v, f)
    base = [str(CLI), "--obj", str(obj), "--env", str(scenes.ASSETS / "Sky_16.png"), "--size", f"{W}x{H}", "--spp", "2", "--frames", str(frames),
            "--cam", "-2,1.5,1.0,-90,0", "--dump-targets"]
    a = subprocess.run(base + ["--out", str(tmp_path / "plain")], capture_output=True, text=True, timeout=300)
    assert a.returncode == 0, a.stdout + a.stderr
    b = subprocess.run(base + ["--ranks", "1", "--gather-every", "2", "--out", str(tmp_path / "ranks")], capture_output=True, text=True, timeout=300)
    assert b.returncode == 0, b.stdout + b.stderr
    assert "[RCCL] 1 ranks, communicator up" in b.stdout and "tile-parallel" in b.stdout
    for suffix in (".png", "_color.pfm", "_motion.pfm", "_gpos.pfm", "_gnrm.pfm"):
        assert (tmp_path / f"plain{suffix}").read_bytes() == (tmp_path / f"ranks{suffix}").read_bytes(), suffix
    assert not glob.glob(str(tmp_path / "ranks.rccl_id*"))


def _gpu_count():
    import torch
    return torch.cuda.device_count()          # counting devices does not initialise the GPU


@pytest.mark.gpu
@pytest.mark.skipif(_gpu_count() < 2, reason="needs two GPUs: RCCL refuses two ranks on one device")
@pytest.mark.parametrize("ranks", [2, 4, 8])
def test_cli_two_or_more_ranks_equal_the_single_process_run(tmp_path, ranks):
    """The real thing on a multi-GPU node: rt_cli --ranks N -- one forked process per GPU, ncclCommInitRank with world N,
    grouped send / recv of the tile blocks to rank 0 after every second frame, all four targets gathered for the present -- writes
    the same bytes as one process on one GPU."""
    if _gpu_count() < ranks:
        pytest.skip(f"{ranks} ranks need {ranks} GPUs")
    W, H, frames = 320, 200, 5
    v, f = rt.meshgen.bunny_standin(3)
    obj = tmp_path / "blob.obj"
    rt.meshgen.write_obj(obj, v, f)
    base = [str(CLI), "--obj", str(obj), "--env", str(scenes.ASSETS / "Sky_16.png"), "--size", f"{W}x{H}", "--spp", "2", "--frames", str(frames),
            "--cam", "-2,1.5,1.0,-90,0", "--dump-targets"]
    a = subprocess.run(base + ["--out", str(tmp_path / "plain")], capture_output=True, text=True, timeout=300)
    assert a.returncode == 0, a.stdout + a.stderr
    b = subprocess.run(base + ["--ranks", str(ranks), "--gather-every", "2", "--out", str(tmp_path / "ranks")], capture_output=True, text=True, timeout=600)
    assert b.returncode == 0, b.stdout + b.stderr
    assert f"[RCCL] {ranks} ranks, communicator up" in b.stdout
    for suffix in (".png", "_color.pfm", "_motion.pfm", "_gpos.pfm", "_gnrm.pfm"):
        assert (tmp_path / f"plain{suffix}").read_bytes() == (tmp_path / f"ranks{suffix}").read_bytes(), suffix

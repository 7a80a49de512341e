"""GPU parity for the BASELINE.json configurations round 1 left unexercised ("configs_untested"):

* configs[4] "64 spp + temporal accumulation over 32 frames": the history weights 0.85 / 0.92 / 0.96 switch at
  uFrameIndex 8 and 32 (shaders/rt/rt_taa.glsl:91-104, include/render/RenderParams.h:189-195) and the COLOR0 ring of
  the frame lanes wraps many times -- 36 consecutive frames against the oracle, with and without a host sync between
  frames, on both pipelines.
* configs[3] "3840x2160, 16 spp": full-size property test (wavefront == megakernel bit for bit, reproducible) plus an
  oracle window, as test_full_size_wavefront_equals_megakernel does for configs[1].
* chunked ray queues at 16 spp with a budget that cuts the frame into >= 8 chunks, and the 4K / 16 spp / 128 MB case
  whose sweep ended round 1's gpurun_out/q.log (491 chunks: more than the then fixed-size cursor table held; the
  renderer refused it with RT_ERR_UNSUPPORTED -- the table now grows).
* counters after rt_reset_counters with frames in flight on several lanes.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

import opengl_raytracing_amd as rt
import scenes

pytestmark = pytest.mark.gpu


def _assert_equal(got, want, orc, what):
    for g, w, n in zip(got, want, ("color", "motion", "gpos", "gnrm")):
        st = orc.compare(g, w)
        assert st["rmse"] < 1e-4 and st["bit_diff"] == 0, f"{what}/{n}: {st}"


@pytest.mark.parametrize("pipeline", ["wave", "mega"])
def test_accumulation_over_36_frames_crosses_the_weight_regimes(orc, pipeline):
    W, H, FRAMES = 80, 48, 36
    nodes, tris = scenes.bunny_bvh(3)
    faces = scenes.tiny_env(8)
    p = rt.default_render_params()
    p.sppPerFrame = 2
    cam = scenes.camera("closeup", aspect=W / H)
    us = [rt.frame_uniforms(p, cam, W, H, f, True, nodes.shape[0], tris.shape[0]) for f in range(FRAMES)]
    want, prev = [], None
    for u in us:
        w, _ = orc.render(u, nodes, tris, faces, prev)
        want.append(w)
        prev = w[0]
    # the three regimes really are in play: the EMA weight changes what frame 8 / 32 do with the same history
    assert us[7].frameIndex == 7 and us[8].frameIndex == 8 and us[32].frameIndex == 32
    pipe = rt.RT_PIPELINE_WAVEFRONT if pipeline == "wave" else rt.RT_PIPELINE_MEGAKERNEL
    with rt.Renderer(pipeline=pipe) as r:
        r.upload_bvh(nodes, tris)
        r.upload_env(faces)
        r.resize(W, H)
        for f, u in enumerate(us):                 # synchronising read after every frame
            r.render_frame(u)
            _assert_equal(r.read_all(), want[f], orc, f"{pipeline} frame {f}")
        assert r.frame_index == FRAMES
        r.reset_accum()
        for u in us:                               # pipelined: frames overlap on the lanes, the ring wraps 9-12 times
            r.render_frame(u)
        _assert_equal(r.read_all(), want[-1], orc, f"{pipeline} pipelined, last frame")


def test_accumulation_over_40_frames_analytic_scene(orc):
    W, H, FRAMES = 64, 48, 40
    faces = scenes.tiny_env(8)
    p = rt.default_render_params()
    cam = scenes.camera("default", aspect=W / H)
    with rt.Renderer() as r:
        r.upload_env(faces)
        r.resize(W, H)
        prev = None
        for f in range(FRAMES):
            u = rt.frame_uniforms(p, cam, W, H, f, False)
            r.render_frame(u)
            want, _ = orc.render(u, env_faces=faces, prev=prev)
            if f in (0, 7, 8, 9, 31, 32, 33, FRAMES - 1):
                _assert_equal(r.read_all(), want, orc, f"analytic frame {f}")
            prev = want[0]


def test_4k_16spp_wavefront_equals_megakernel(orc):
    """BASELINE configs[3] on one GPU: 3840x2160, 16 spp (the 8-GPU version tiles exactly this frame)."""
    W, H = 3840, 2160
    nodes, tris = scenes.bunny_bvh(6)
    faces = scenes.env_faces("Sky_01")
    p = rt.default_render_params()
    p.sppPerFrame = 16
    cam = scenes.camera("closeup")
    outs = {}
    for name, pipe in (("mega", rt.RT_PIPELINE_MEGAKERNEL), ("wave", rt.RT_PIPELINE_WAVEFRONT)):
        with rt.Renderer(pipeline=pipe) as r:
            r.upload_bvh(nodes, tris)
            r.upload_env(faces)
            r.resize(W, H)
            for frame in range(2):
                r.render_frame(rt.frame_uniforms(p, cam, W, H, frame, True, nodes.shape[0], tris.shape[0]))
            outs[name] = r.read_all()
            if name == "wave":
                tr = r.traced_rays()
                assert tr.frames == 2 and tr.hitPixels > W * H // 2     # 2 frames x ~45 % of the pixels
    for x, y in zip(outs["mega"], outs["wave"]):
        assert np.array_equal(x, y)
    with rt.Renderer() as r:
        r.upload_bvh(nodes, tris)
        r.upload_env(faces)
        r.resize(W, H)
        u = rt.frame_uniforms(p, cam, W, H, 0, True, nodes.shape[0], tris.shape[0])
        r.render_frame(u)
        got = r.read_all()
    x0, y0, x1, y1 = 1800, 1000, 1832, 1016
    want, _ = orc.render(u, nodes, tris, faces, None, region=(x0, y0, x1, y1))
    for g, w in zip(got, want):
        assert np.array_equal(g[y0:y1, x0:x1], w[y0:y1, x0:x1])


_CHUNK_CODE = r'''
import sys, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import opengl_raytracing_amd as rt, oracle as orc, scenes
W, H, SPP, SUB, ORACLE = (int(v) for v in sys.argv[1:6])
nodes, tris = scenes.bunny_bvh(SUB); faces = scenes.tiny_env(16)
p = rt.default_render_params(); p.sppPerFrame = SPP
cam = scenes.camera("closeup", aspect=W / H)
with rt.Renderer(pipeline=rt.RT_PIPELINE_WAVEFRONT) as r:
    r.upload_bvh(nodes, tris); r.upload_env(faces); r.resize(W, H)
    prev = None
    for f in range(2):
        u = rt.frame_uniforms(p, cam, W, H, f, True, nodes.shape[0], tris.shape[0])
        r.render_frame(u)
        got = r.read_all()
        if ORACLE:
            want, _ = orc.render(u, nodes, tris, faces, prev, nthreads=16)
            for g, w in zip(got, want):
                assert np.array_equal(g, w)
            prev = want[0]
    np.save(sys.argv[6], np.concatenate([g.reshape(-1) for g in got]))
print("CHUNKED-OK")
'''


def _run_chunked(budget_mb, w, h, spp, subdiv, oracle, out):
    env = dict(os.environ, RT_QUEUE_BUDGET_MB=str(budget_mb))
    r = subprocess.run([sys.executable, "-c", _CHUNK_CODE, str(w), str(h), str(spp), str(subdiv), str(int(oracle)), str(out)],
                       cwd=str(scenes.ROOT), env=env, capture_output=True, text=True, timeout=900)
    assert "CHUNKED-OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_chunked_queues_16spp_ten_chunks(tmp_path):
    """256x160 at 16 spp with a 1 MB budget: 4096-hit chunks, 40960 pixel slots -> 10 chunks; against the oracle."""
    _run_chunked(1, 256, 160, 16, 4, True, tmp_path / "a.npy")


def test_chunked_queues_4k_16spp_128mb_equals_unchunked(tmp_path):
    """The case round 1's budget sweep stopped at: 4K, 16 spp, 128 MB -> 491 chunks (1474 trace launches per frame)."""
    _run_chunked(128, 3840, 2160, 16, 5, False, tmp_path / "small.npy")
    _run_chunked(8192, 3840, 2160, 16, 5, False, tmp_path / "big.npy")
    assert np.array_equal(np.load(tmp_path / "small.npy"), np.load(tmp_path / "big.npy"))


def test_reset_counters_with_frames_in_flight(orc):
    """N pipelined frames on all lanes, rt_reset_counters, one frame: the counters are that one frame's."""
    W, H = 320, 200
    nodes, tris = scenes.bunny_bvh(4)
    p = rt.default_render_params()
    p.sppPerFrame = 2
    cam = scenes.camera("closeup", aspect=W / H)
    with rt.Renderer(pipeline=rt.RT_PIPELINE_MEGAKERNEL, count_work=True) as r:
        r.upload_bvh(nodes, tris)
        r.resize(W, H)
        us = [rt.frame_uniforms(p, cam, W, H, f, True, nodes.shape[0], tris.shape[0]) for f in range(8)]
        for u in us[:7]:
            r.render_frame(u)          # no sync: up to nLanes frames are still running
        r.reset_counters()
        r.render_frame(us[7])
        got = r.counters()
        r.reset_accum()
        r.reset_counters()
        for u in us[:7]:
            r.render_frame(u)
        r.synchronize()
        r.reset_counters()
        r.render_frame(us[7])
        want = r.counters()
    assert got.to_dict() == want.to_dict()
    assert got.rays > W * H


@pytest.mark.parametrize("world,rank", [(1, 0), (3, 1)])
def test_batched_frames_equal_frame_by_frame(orc, world, rank):
    """rt_render_frames: up to 16 consecutive frames of a static camera in one set of launches (virtual tiles per frame of the batch,
    per-frame jitter and frame index, history chained through fp16 in the resolve kernel) == the same frames one by one, bit for
    bit, for every batch length and in the middle of an accumulation; a moving frame or a change of any other uniform ends a batch."""
    W, H = 200, 120
    nodes, tris = scenes.bunny_bvh(4)
    faces = scenes.tiny_env(16)
    p = rt.default_render_params()
    p.sppPerFrame = 2
    cam = scenes.camera("closeup", aspect=W / H)
    us = [rt.frame_uniforms(p, cam, W, H, f, True, nodes.shape[0], tris.shape[0]) for f in range(40)]
    assert us[3].jitter[0] != us[4].jitter[0]              # the frames of a batch really differ in their jitter
    with rt.Renderer(rank=rank, world_size=world) as one, rt.Renderer(rank=rank, world_size=world) as many:
        for r in (one, many):
            r.upload_bvh(nodes, tris)
            r.upload_env(faces)
            r.resize(W, H)
        f = 0
        for n in (1, 2, 3, 8, 5, 4, 11):                   # 11 = one batch of 8 + one of 3
            for u in us[f:f + n]:
                one.render_frame(u)
            many.render_frames(us[f:f + n])
            f += n
            assert one.frame_index == many.frame_index == f
            for a, b in zip(one.read_all(), many.read_all()):
                assert np.array_equal(a, b), (n, f)
        if world == 1:                                      # the last frame against the oracle as well (34 frames deep)
            prev = None
            for u in us[:f]:
                want, _ = orc.render(u, nodes, tris, faces, prev)
                prev = want[0]
            for a, b in zip(many.read_all(), want):
                assert np.array_equal(a, b)
        tr = many.traced_rays()
        assert tr.frames == f
        # a different spp in the middle: the batch is cut there, the result is still the sequential one
        p2 = rt.default_render_params()
        p2.sppPerFrame = 3
        mixed = [us[f], us[f + 1], rt.frame_uniforms(p2, cam, W, H, f + 2, True, nodes.shape[0], tris.shape[0]), us[f + 3]]
        for u in mixed:
            one.render_frame(u)
        many.render_frames(mixed)
        for a, b in zip(one.read_all(), many.read_all()):
            assert np.array_equal(a, b)


def test_batched_frames_fall_back_for_the_analytic_scene_and_the_megakernel(orc):
    W, H = 96, 64
    nodes, tris = scenes.bunny_bvh(3)
    p = rt.default_render_params()
    cam = scenes.camera("default", aspect=W / H)
    for pipe, use_bvh in ((rt.RT_PIPELINE_AUTO, False), (rt.RT_PIPELINE_MEGAKERNEL, True)):
        us = [rt.frame_uniforms(p, cam, W, H, f, use_bvh, nodes.shape[0], tris.shape[0]) for f in range(5)]
        with rt.Renderer(pipeline=pipe) as a, rt.Renderer(pipeline=pipe) as b:
            for r in (a, b):
                r.upload_bvh(nodes, tris)
                r.resize(W, H)
            for u in us:
                a.render_frame(u)
            b.render_frames(us)
            for x, y in zip(a.read_all(), b.read_all()):
                assert np.array_equal(x, y)


@pytest.mark.parametrize("budget_mb", [None, 6144])
def test_1080p_batches_of_eight_equal_frame_by_frame_and_the_oracle(orc, monkeypatch, budget_mb):
    """The mode bench.py times: 1920x1080, 4 spp, 81 920 triangles, rt_render_frames in batches of 8 -- 17 frames = two
    full batches and a remainder of one -- against one rt_render_frame per frame: all four targets bit-equal after every batch; and a
    64x32 window of the last frame (16 frames of history deep) against the oracle.  With the default ray-queue budget a batch is one
    chunk; with 6 GB it is cut into three (hit count read back, hits dealt over equal chunks)."""
    if budget_mb:
        monkeypatch.setenv("RT_QUEUE_BUDGET_MB", str(budget_mb))     # read when the renderer is created
    W, H, FRAMES = 1920, 1080, 17
    nodes, tris = scenes.bunny_bvh(6)
    faces = scenes.env_faces("Sky_01")
    p = rt.default_render_params()
    p.sppPerFrame = 4
    cam = scenes.camera("closeup")
    us = [rt.frame_uniforms(p, cam, W, H, f, True, nodes.shape[0], tris.shape[0]) for f in range(FRAMES)]
    with rt.Renderer() as one, rt.Renderer() as many:
        for r in (one, many):
            r.upload_bvh(nodes, tris)
            r.upload_env(faces)
            r.resize(W, H)
        f = 0
        for n in (8, 8, 1):
            for u in us[f:f + n]:
                one.render_frame(u)
            many.render_frames(us[f:f + n])
            f += n
            for a, b in zip(one.read_all(), many.read_all()):
                assert np.array_equal(a, b), f
        got = many.read_all()
        tr = many.traced_rays()
        assert tr.frames == FRAMES and tr.hitPixels > FRAMES * W * H // 4
    x0, y0, x1, y1 = 900, 500, 964, 532
    prev = None
    for u in us:
        want, _ = orc.render(u, nodes, tris, faces, prev, region=(x0, y0, x1, y1), nthreads=16)
        prev = want[0]
    for g, w in zip(got, want):
        assert np.array_equal(g[y0:y1, x0:x1], w[y0:y1, x0:x1])


def test_1080p_whole_frames_against_the_oracle(orc):
    """configs[1] at full size against the oracle itself, every pixel of every target (full-size oracle checks were a
    64x32 window): 1920x1080, 4 spp, 81 920 triangles, three accumulating frames submitted as one batch; the oracle renders the same three
    whole frames on the box's 16 host threads (about 6 s each).  The hit-pixel count is the oracle's too."""
    W, H, FRAMES = 1920, 1080, 3
    nodes, tris = scenes.bunny_bvh(6)
    faces = scenes.env_faces("Sky_01")
    p = rt.default_render_params()
    p.sppPerFrame = 4
    cam = scenes.camera("closeup")
    us = [rt.frame_uniforms(p, cam, W, H, f, True, nodes.shape[0], tris.shape[0]) for f in range(FRAMES)]
    with rt.Renderer() as ren:
        ren.upload_bvh(nodes, tris)
        ren.upload_env(faces)
        ren.resize(W, H)
        ren.render_frames(us)
        got = ren.read_all()
        hit_pixels = ren.traced_rays().hitPixels
    prev, hits = None, 0
    for u in us:
        want, cnt = orc.render(u, nodes, tris, faces, prev, nthreads=16)
        prev = want[0]
        hits += cnt.hitPixels
    for g, w in zip(got, want):
        assert np.array_equal(g, w)
    assert hit_pixels == hits


def test_16spp_whole_frame_and_4k_band_against_the_oracle(orc):
    """configs[2] run A (1920x1080, 16 spp) against the oracle over the WHOLE first frame, and configs[3] (3840x2160, 16 spp) over a band of 256 full-width
    rows through the mesh -- every target, bit for bit (the full-size oracle checks were 64x32 / 32x16 windows; the rest of those frames
    was covered by wavefront == megakernel only).  About 35 s of oracle time on the box's 16 host threads."""
    nodes, tris = scenes.bunny_bvh(6)
    faces = scenes.env_faces("Sky_01")
    p = rt.default_render_params()
    p.sppPerFrame = 16
    cam = scenes.camera("closeup")
    for (W, H), region in (((1920, 1080), None), ((3840, 2160), (0, 952, 3840, 1208))):
        u = rt.frame_uniforms(p, cam, W, H, 0, True, nodes.shape[0], tris.shape[0])
        with rt.Renderer() as r:
            r.upload_bvh(nodes, tris)
            r.upload_env(faces)
            r.resize(W, H)
            r.render_frame(u)
            got = r.read_all()
        want, cnt = orc.render(u, nodes, tris, faces, None, region=region, nthreads=16)
        x0, y0, x1, y1 = region if region else (0, 0, W, H)
        assert cnt.hitPixels > (x1 - x0) * (y1 - y0) // 4          # the band / frame really crosses the mesh
        for g, w in zip(got, want):
            assert np.array_equal(g[y0:y1, x0:x1], w[y0:y1, x0:x1]), (W, H)


_BATCH_CHUNK_CODE = r'''
import sys, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import opengl_raytracing_amd as rt, oracle as orc, scenes
W, H, SPP, SUB = 256, 160, 2, 4
nodes, tris = scenes.bunny_bvh(SUB); faces = scenes.tiny_env(16)
p = rt.default_render_params(); p.sppPerFrame = SPP
cam = scenes.camera("closeup", aspect=W / H)
us = [rt.frame_uniforms(p, cam, W, H, f, True, nodes.shape[0], tris.shape[0]) for f in range(9)]
with rt.Renderer(pipeline=rt.RT_PIPELINE_WAVEFRONT) as r:
    r.upload_bvh(nodes, tris); r.upload_env(faces); r.resize(W, H)
    r.render_frames(us[:5])          # one batch of five frames: ~90 k hits in 4096-hit chunks, chunk boundaries fall inside frames
    r.render_frames(us[5:])          # and a batch of four
    got = r.read_all()
    hits = r.traced_rays().hitPixels
assert hits > 9 * 4096 * 2, hits     # really many chunks per batch
prev = None
for u in us:
    want, _ = orc.render(u, nodes, tris, faces, prev, nthreads=16)
    prev = want[0]
for g, w in zip(got, want):
    assert np.array_equal(g, w)
print("BATCH-CHUNKED-OK")
'''


@pytest.mark.parametrize("from_slots", [0, 1])
def test_batched_frames_with_chunked_queues_against_the_oracle(from_slots):
    """rt_render_frames x chunked ray queues (RT_QUEUE_BUDGET_MB=1: 4096-hit chunks), 256x160 at 2 spp, batches of 5 and 4 frames, against
    the oracle 9 frames deep.  from_slots=1 launches the chunk loop for every pixel slot as rounds 1-2 did (RT_CHUNKS_FROM_SLOTS): more than
    half of the launch sets of a batch are then empty trailing chunks; the default reads the hit count back and launches none."""
    env = dict(os.environ, RT_QUEUE_BUDGET_MB="1", RT_CHUNKS_FROM_SLOTS=str(from_slots))
    r = subprocess.run([sys.executable, "-c", _BATCH_CHUNK_CODE], cwd=str(scenes.ROOT), env=env, capture_output=True, text=True, timeout=900)
    assert "BATCH-CHUNKED-OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize("cam_kind", ["default", "closeup"])
def test_million_triangle_scene_whole_1080p_frames_against_the_oracle(orc, cam_kind):
    """The 1 M-triangle scene of configs[4] at 1920x1080 / 4 spp, two accumulating frames as one batch, against the oracle over the whole frame and every
    target -- the scene whose any-hit launches walk the QUANTISED nodes by default (any-hit tree beyond 4 MB), here at full size and not
    only at 640x360; the same frames with the exact nodes forced (RT_QNODES=0 is read at upload) must be the same bits."""
    v, f = rt.meshgen.million_triangle_scene()
    nodes, tris = rt.build_bvh(rt.gather_triangles(v, f, np.eye(4, dtype=np.float32).reshape(-1)))
    faces = scenes.env_faces("Sky_01")
    W, H = 1920, 1080
    p = rt.default_render_params()
    p.sppPerFrame = 4
    cam = scenes.camera(cam_kind)
    us = [rt.frame_uniforms(p, cam, W, H, f_, True, nodes.shape[0], tris.shape[0]) for f_ in range(2)]
    with rt.Renderer() as r:
        r.upload_bvh(nodes, tris)
        info = r.scene_info()
        if "RT_QNODES" not in os.environ:
            assert info.bytesNodes4 == info.nWide4 * 64 + 131072 * 32   # the quantised form is the one that is walked: 64 B per node + 32 B per leaf
        r.upload_env(faces)
        r.resize(W, H)
        r.render_frames(us)
        got = r.read_all()
    prev = None
    for u in us:
        want, cnt = orc.render(u, nodes, tris, faces, prev, nthreads=16)
        prev = want[0]
    assert cnt.hitPixels > W * H // 16         # a tenth of the frame and more shows the spheres
    for g, w_ in zip(got, want):
        assert np.array_equal(g, w_)


def test_config4_scene_spp_and_accumulation_together(orc):
    """BASELINE configs[4] with its three dimensions TOGETHER (they had only run one at a time): the 1 M-triangle multi-object
    scene, 64 spp, temporal accumulation over 32 frames -- at 192x120 (the oracle's cost is per pixel, sample and frame), rendered in batches of eight
    through rt_render_frames (32 frames = four batches; the history weight switches at frames 8 and 32; one launch set = 8 x 64 samples per hit, several
    chunks of the default queue budget are NOT needed at this size, the deep tree's 24- / 36-entry stacks are) -- against the oracle on a 24x12 window
    with the whole 32-frame history chain, and wavefront == frame-by-frame rendering on all four full targets."""
    v, f = rt.meshgen.million_triangle_scene()
    nodes, tris = rt.build_bvh(rt.gather_triangles(v, f, np.eye(4, dtype=np.float32).reshape(-1)))
    assert tris.shape[0] == 1_000_000
    faces = scenes.tiny_env(16)
    W, H, FRAMES = 192, 120, 32
    p = rt.default_render_params()
    p.sppPerFrame = 64
    cam = scenes.camera("default", aspect=W / H)
    us = [rt.frame_uniforms(p, cam, W, H, f_, True, nodes.shape[0], tris.shape[0]) for f_ in range(FRAMES)]
    with rt.Renderer() as many, rt.Renderer() as one:
        for r in (many, one):
            r.upload_bvh(nodes, tris)
            r.upload_env(faces)
            r.resize(W, H)
        for b in range(0, FRAMES, 8):
            many.render_frames(us[b:b + 8])
        for u in us:
            one.render_frame(u)
        got = many.read_all()
        for a, b in zip(got, one.read_all()):
            assert np.array_equal(a, b)
        assert many.traced_rays().hitPixels > FRAMES * 1000
    x0, y0 = 84, 54
    x1, y1 = x0 + 24, y0 + 12
    prev = None
    for u in us:
        want, _ = orc.render(u, nodes, tris, faces, prev, region=(x0, y0, x1, y1), nthreads=16)
        prev = want[0]
    assert want[2][y0:y1, x0:x1].any()          # the window shows geometry
    for g, w_ in zip(got, want):
        assert np.array_equal(g[y0:y1, x0:x1], w_[y0:y1, x0:x1])


def test_config4_at_its_own_size(orc):
    """BASELINE configs[4] AS WRITTEN on one GPU: the 1 M-triangle multi-object scene, 1920x1080, 64 spp, temporal accumulation over frames
    0..31 -- the run DESIGN.md times at ~250 ms per frame and nobody had looked at.  Rendered in batches of eight through rt_render_frames (each batch's ray
    queues are several chunks of the 16 GB budget; the any-hit launches walk the quantised nodes) and compared with
      (i)  the same 32 frames rendered one rt_render_frame at a time by a second context: all four targets, every pixel, bit for bit;
      (ii) the oracle on a 32 x 16 window through geometry with the whole 32-frame history chain (64 spp x 512 pixels x 32 frames = 1 M oracle samples; the
           weights of rt_taa.glsl:91-104 switch at frames 8 and 32): bit for bit.
    About 20 s of GPU time and 10 s of host time (scene + tree build)."""
    v, f = rt.meshgen.million_triangle_scene()
    nodes, tris = rt.build_bvh(rt.gather_triangles(v, f, np.eye(4, dtype=np.float32).reshape(-1)))
    assert tris.shape[0] == 1_000_000
    faces = scenes.env_faces("Sky_01")
    W, H, FRAMES = 1920, 1080, 32
    p = rt.default_render_params()
    p.sppPerFrame = 64
    cam = scenes.camera("default")
    us = [rt.frame_uniforms(p, cam, W, H, f_, True, nodes.shape[0], tris.shape[0]) for f_ in range(FRAMES)]
    with rt.Renderer() as many:
        many.upload_bvh(nodes, tris)
        many.upload_env(faces)
        many.resize(W, H)
        for b in range(0, FRAMES, 8):
            many.render_frames(us[b:b + 8])
        got = many.read_all()
        hit_pixels = many.traced_rays().hitPixels
    assert hit_pixels > FRAMES * W * H // 16
    with rt.Renderer() as one:                   # (after the first context is gone: each takes tens of GB of ray queues at this size)
        one.upload_bvh(nodes, tris)
        one.upload_env(faces)
        one.resize(W, H)
        for u in us:
            one.render_frame(u)
        for a, b in zip(got, one.read_all()):
            assert np.array_equal(a, b)
    # a window that shows geometry: the row band and column run with the most hit pixels in the position target
    geo = got[2][..., 3] != 0
    y0 = int(np.argmax(np.convolve(geo.sum(axis=1), np.ones(16, int), "valid")))
    x0 = int(np.argmax(np.convolve(geo[y0:y0 + 16].sum(axis=0), np.ones(32, int), "valid")))
    x1, y1 = x0 + 32, y0 + 16
    assert geo[y0:y1, x0:x1].sum() > 256
    prev = None
    for u in us:
        want, _ = orc.render(u, nodes, tris, faces, prev, region=(x0, y0, x1, y1), nthreads=16)
        prev = want[0]
    for g, w_ in zip(got, want):
        assert np.array_equal(g[y0:y1, x0:x1], w_[y0:y1, x0:x1])


def test_shared_ray_arenas_and_memory_info(orc):
    """Round 4: the ray-queue arenas are shared by the frame lanes (two for four lanes; a launch set of several chunks takes its lane's own) and
    rt_get_memory_info reports them.  Twelve batches of four frames keep all four lanes and both arenas turning over; the last frame equals the
    oracle's (11 batches of history), and the report is consistent: two arenas, four lanes, device figures from hipMemGetInfo."""
    W, H, K, NB = 160, 96, 4, 12
    nodes, tris = scenes.bunny_bvh(3)
    faces = scenes.tiny_env(8)
    p = rt.default_render_params()
    p.sppPerFrame = 2
    cam = scenes.camera("closeup", aspect=W / H)
    us = [rt.frame_uniforms(p, cam, W, H, f, True, nodes.shape[0], tris.shape[0]) for f in range(K * NB)]
    with rt.Renderer(pipeline=rt.RT_PIPELINE_WAVEFRONT) as r:
        r.upload_bvh(nodes, tris)
        r.upload_env(faces)
        r.resize(W, H)
        for b in range(NB):
            r.render_frames(us[b * K:(b + 1) * K])      # no synchronisation in between: lanes and arenas overlap
        got = r.read_all()
        m = r.memory_info()
    assert m.queueArenaBytes > 0 and m.frameArrayBytes > 0 and m.hybridArenaBytes == 0 and 1 <= m.queueArenas <= m.lanes
    assert 0 < m.deviceFreeBytes < m.deviceTotalBytes
    if not any(k in os.environ for k in ("RT_LANES", "RT_ARENAS", "RT_QUEUE_BUDGET_MB", "RT_CHUNKS_FROM_SLOTS")):   # (tools/r04_stress.sh overrides them)
        assert m.lanes == 4 and m.queueArenas == 2
    prev = None
    for u in us:
        want, _ = orc.render(u, nodes, tris, faces, prev, nthreads=16)
        prev = want[0]
    for g, w_ in zip(got, want):
        assert np.array_equal(g, w_)


def test_mixed_sequences_fall_back_per_run(orc):
    """rt_render_frames decides batching per run of frames, not once from the first frame -- [BVH, analytic, analytic, BVH, BVH]
    renders (the analytic frames one by one on the megakernel) instead of failing with "a frame batch reached the megakernel" after frame 0."""
    W, H = 96, 64
    nodes, tris = scenes.bunny_bvh(3)
    faces = scenes.tiny_env(8)
    p = rt.default_render_params()
    cam = scenes.camera("closeup", aspect=W / H)
    kinds = [True, False, False, True, True, rt.RT_SCENE_HYBRID, rt.RT_SCENE_HYBRID, True]
    us = [rt.frame_uniforms(p, cam, W, H, f, k, nodes.shape[0], tris.shape[0]) for f, k in enumerate(kinds)]
    with rt.Renderer() as a, rt.Renderer() as b:
        for r in (a, b):
            r.upload_bvh(nodes, tris)
            r.upload_env(faces)
            r.resize(W, H)
        for u in us:
            a.render_frame(u)
        b.render_frames(us)
        assert a.frame_index == b.frame_index == len(us)
        for x, y in zip(a.read_all(), b.read_all()):
            assert np.array_equal(x, y)


@pytest.mark.parametrize("thresh", [0.0, -1.0])
def test_batching_is_off_when_the_still_branch_cannot_be_taken(orc, thresh):
    """with uTaaStillThresh <= 0 resolveTAA never takes the still branch (length(0,0) < 0 is false), so frame k of a batch would
    reproject from the history texture of BEFORE the batch.  rt_render_frames renders such frames one by one; the result is the sequential
    one and the oracle's."""
    W, H = 120, 72
    nodes, tris = scenes.bunny_bvh(3)
    faces = scenes.tiny_env(8)
    p = rt.default_render_params()
    p.sppPerFrame = 2
    p.taaStillThresh = thresh
    cam = scenes.camera("closeup", aspect=W / H)
    us = [rt.frame_uniforms(p, cam, W, H, f, True, nodes.shape[0], tris.shape[0]) for f in range(6)]
    with rt.Renderer() as a, rt.Renderer() as b:
        for r in (a, b):
            r.upload_bvh(nodes, tris)
            r.upload_env(faces)
            r.resize(W, H)
        for u in us:
            a.render_frame(u)
        b.render_frames(us)
        for x, y in zip(a.read_all(), b.read_all()):
            assert np.array_equal(x, y)
        got = b.read_all()
    prev = None
    for u in us:
        want, _ = orc.render(u, nodes, tris, faces, prev)
        prev = want[0]
    _assert_equal(got, want, orc, f"taaStillThresh={thresh}")

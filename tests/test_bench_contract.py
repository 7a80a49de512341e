"""bench.py's output contract, on a small variant of the workload (the driver runs the default line itself): ONE JSON line with the
metric keys, a physical roofline (0 < frac <= 1 against the HBM peak AND against the L1 access ceiling beside it), the launch duration
from real launches, the batched-vs-frame-by-frame self-check, the CPU baseline with one thread and all cores from the -O3 -march=native
oracle build, and value / value_traversed consistent with ms_per_step.  Also: a supplied .obj drives the same line (--obj), and the
self-launcher brings up the tile-parallel code path (process group, RCCL communicator, gather) with a world of one."""
import json
import subprocess
import sys

import pytest

import opengl_raytracing_amd as rt
import scenes

pytestmark = pytest.mark.gpu

SMALL = ["--steps", "6", "--warmup", "2", "--size", "640x360", "--subdiv", "4"]


def _bench(args, timeout=600):
    out = subprocess.run([sys.executable, "bench.py"] + args, cwd=str(scenes.ROOT), capture_output=True, text=True, timeout=timeout)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    return json.loads(lines[0])


def test_bench_line_contract():
    d = _bench(SMALL + ["--cpu-seconds", "0.5"])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "cpu_baseline", "value_traversed"):
        assert k in d, k
    assert d["unit"] == "Mray/s" and d["n_gpus"] == 1 and d["steps"] == 6 and d["warmup"] == 2 and d["vs_baseline"] is None and d["dtype"] == "f32"
    assert "workload" in d["config"] and "model" not in d["config"]
    rays = d["config"]["rays_per_frame"]
    assert abs(d["value"] - rays / (d["ms_per_step"] * 1e-3) / 1e6) / d["value"] < 0.02
    assert 0 < d["value_traversed"] < d["value"]
    # the timed mode (batches of frames) is checked against frame-by-frame rendering inside the run
    c = d["config"]
    assert c["batched_equals_frame_by_frame"] is True and c["color0_sha256"] == c["color0_sha256_frame_by_frame"] and len(c["color0_sha256"]) == 64
    assert c["ms_per_step_frame_by_frame"] > 0
    assert c["stage_events_in_timed_region"] is False and c["stage_frames"] >= 1      # one GPU: the stage spans come from an untimed pass
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert 0 < r["frac"] <= 1 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert r["algorithmic_bytes_per_launch"] > 0 and r["avg_launch_ms"] > 0 and r["launches"] >= 1 and r["frames_per_launch"] >= 1
    assert "ONE launch set in flight" in r["avg_launch_ms_source"]
    assert r["traffic"] is None or r["traffic_source"]["kind"] == "profiled_offline"
    assert r["reference_layout"] is None or r["reference_layout"]["frac"] is None
    if r["kernel"].startswith("trace_"):
        l1 = r["l1_gather"]
        assert l1 is not None and 0 < l1["merge_factor"] <= 1 and 0 < l1["frac"] <= 1        # a ceiling that is exceeded is not a ceiling
        assert abs(l1["frac"] - l1["achieved"] / l1["peak"]) < 1e-9 and l1["peak"] == 256 * 2.4
    # the metric's RMSE leg: the last timed frame against the oracle, a window around the frame centre, whole history chain
    pz = d["parity"]
    assert pz["vs"] == "oracle" and pz["ok"] is True and pz["rmse"] < 1e-4 and pz["bit_diff"] == 0 and pz["outliers_gt_1e-2"] == 0
    assert pz["frames_chained"] == 8 and pz["frame"] == 7 and set(pz["targets"]) == {"color", "motion", "gpos", "gnrm"}
    x0, y0, x1, y1 = pz["window"]
    assert (x1 - x0, y1 - y0) == (256, 128) and x0 <= 320 < x1 and y0 <= 180 < y1
    b = d["cpu_baseline"]
    assert b["kind"] == "port" and b["cores"] >= 1 and b["value"] > 0 and b["single_thread"]["cores"] == 1 and b["single_thread"]["value"] > 0
    assert "-O3 -march=native" in b["sample"]


def test_bench_obj_line_equals_the_standin_line(tmp_path):
    """bench.py --obj: the mesh written as .obj and read back by rt_load_obj gives the same frames (COLOR0 hash), rays and hit pixels as the
    generated mesh -- a supplied Stanford bunny would drive the headline line the same way (application.cpp:260-272)."""
    v, f = rt.meshgen.bunny_standin(4)
    obj = tmp_path / "standin.obj"
    rt.meshgen.write_obj(obj, v, f)
    common = SMALL + ["--cpu-seconds", "0", "--no-default-camera", "--no-diagnostics"]
    a = _bench(common)
    b = _bench(common + ["--obj", str(obj)])
    for k in ("rays_per_frame", "hit_pixels", "rays_traversed_per_frame", "color0_sha256"):
        assert a["config"][k] == b["config"][k], k
    assert "standin.obj (5120 tris" in b["config"]["workload"] and b["data"] == "supplied .obj"
    assert b["config"]["batched_equals_frame_by_frame"] is True


def test_bench_self_launch_rehearsal_world_of_one():
    """--launch --force-gather: bench.py starts torch.distributed.run itself (as it does for --gpus N > 1), the rank brings up the process group
    and the library's RCCL communicator and gathers after every batch."""
    d = _bench(SMALL + ["--cpu-seconds", "0", "--no-default-camera", "--launch", "--force-gather", "--gpus", "1"])
    assert d["n_gpus"] == 1 and d["config"]["gather"]["path"].startswith("library-owned RCCL")
    assert d["config"]["batched_equals_frame_by_frame"] is True
    # the self-diagnosing N-GPU block, here for a world of one over a real RCCL communicator
    m = d["config"]["multi_gpu"]
    assert m["rccl_world"] == [1] and m["rccl_rank"] == [0] and m["fallback"] == [False] and m["device"] == [0]
    assert len(m["per_rank_ms"]["values"]) == 1 and 0 < m["per_rank_ms"]["max"] <= d["ms_per_step"] * 1.05 and abs(m["imbalance"] - 1.0) < 1e-9
    assert m["gathers"] >= 1 and m["gather_ms_per_batch"] > 0 and m["gather_bytes"] == 0     # a world of one receives nothing
    assert m["efficiency_vs_n1"] is None
    # parity of the ASSEMBLED frame (the gather is inside the comparison)
    assert d["parity"]["ok"] is True and d["parity"]["bit_diff"] == 0 and list(d["parity"]["targets"]) == ["color"]


@pytest.mark.parametrize("ranks", [2, 3])
def test_bench_ranks_as_processes_on_one_gpu(ranks):
    """--gpus N --rehearse-one-gpu: bench.py starts N rank processes itself (the form the driver uses), every rank renders its tiles on GPU 0,
    the group is gloo and the gathers are staged through the host (RCCL refuses two ranks on one device).  What this covers that no
    one-process test can: the launch, RANK / WORLD_SIZE handling, barriers and max-over-ranks timing, counters summed over ranks, one gather
    per batch from N processes -- and rank 0 compares the assembled frame with the one a single context renders, bit for bit."""
    d = _bench(SMALL + ["--cpu-seconds", "0", "--no-default-camera", "--no-diagnostics", "--gpus", str(ranks), "--rehearse-one-gpu", "--n1-ms", "1.0"], timeout=900)
    assert d["n_gpus"] == ranks and d["scaling"] == "strong"
    c = d["config"]
    assert "gloo" in c["gather"]["path"] and "error" not in c["gather"]
    assert c["batched_equals_frame_by_frame"] is True        # every rank: its own tiles
    assert c["assembled_equals_single_rank"] is True
    m = c["multi_gpu"]
    assert len(m["per_rank_ms"]["values"]) == ranks and m["per_rank_ms"]["min"] > 0 and m["imbalance"] >= 1.0
    assert m["rccl_world"] == [-1] * ranks and m["fallback"] == [False] * ranks       # the rehearsal runs over gloo, and says so
    assert m["gathers"] >= 1 and m["gather_ms_per_batch"] > 0
    tiles_other = m["gather_bytes"]                                                   # what rank 0 receives per gather: the other ranks' blocks
    assert tiles_other > 0 and tiles_other % (256 * 8) == 0
    assert abs(m["efficiency_vs_n1"] - 1.0 / (d["ms_per_step"] * ranks)) < 1e-9         # --n1-ms 1.0
    assert d["parity"]["ok"] is True and d["parity"]["bit_diff"] == 0                   # the assembled frame against the oracle
    one = _bench(SMALL + ["--cpu-seconds", "0", "--no-default-camera", "--no-diagnostics"])
    assert c["rays_per_frame"] == one["config"]["rays_per_frame"]   # the ranks' reference-unit counters add up to the whole frame's
    assert c["hit_pixels"] == one["config"]["hit_pixels"]

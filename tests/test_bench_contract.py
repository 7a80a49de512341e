"""bench.py's output contract, on a small variant of the workload (the driver runs the default line itself): ONE JSON line with the
metric keys, a physical roofline (0 < frac <= 1 against the HBM peak, the binding L1 figure beside it), the CPU baseline with one
thread and all cores from the -O3 -march=native oracle build, and value / value_traversed consistent with ms_per_step."""
import json
import subprocess
import sys

import pytest

import scenes

pytestmark = pytest.mark.gpu


def test_bench_line_contract():
    cmd = [sys.executable, "bench.py", "--steps", "6", "--warmup", "2", "--size", "640x360", "--subdiv", "4", "--cpu-seconds", "0.5"]
    out = subprocess.run(cmd, cwd=str(scenes.ROOT), capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "cpu_baseline", "value_traversed"):
        assert k in d, k
    assert d["unit"] == "Mray/s" and d["n_gpus"] == 1 and d["steps"] == 6 and d["warmup"] == 2 and d["vs_baseline"] is None and d["dtype"] == "f32"
    assert "workload" in d["config"] and "model" not in d["config"]
    rays = d["config"]["rays_per_frame"]
    assert abs(d["value"] - rays / (d["ms_per_step"] * 1e-3) / 1e6) / d["value"] < 0.02
    assert 0 < d["value_traversed"] < d["value"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert 0 < r["frac"] <= 1 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert r["algorithmic_bytes_per_launch"] > 0 and r["avg_launch_ms"] > 0
    assert r["traffic"] is None or r["traffic_source"]["kind"] == "profiled_offline"
    assert r["reference_layout"] is None or r["reference_layout"]["frac"] is None
    if r["l1_gather"]:
        assert 0 < r["l1_gather"]["frac"] < 2
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["single_thread"]["cores"] == 1 and c["single_thread"]["value"] > 0
    assert "-O3 -march=native" in c["sample"]

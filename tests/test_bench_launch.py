"""`python bench.py --gpus N` starts its own ranks (the driver calls it without a launcher).  CPU-only: the launch plan
(--dry-launch), and a real self-launch with a stand-in child that shows the relay / failure / timeout handling.  No GPU is touched:
bench.py decides to launch before it imports torch or the library."""
import json
import os
import subprocess
import sys

import scenes

BENCH = str(scenes.ROOT / "bench.py")


def _run(args, env=None, timeout=120):
    e = dict(os.environ)
    e.pop("WORLD_SIZE", None)
    e.pop("RANK", None)
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH] + args, cwd=str(scenes.ROOT), env=e, capture_output=True, text=True, timeout=timeout)


def test_dry_launch_prints_the_child_command():
    out = _run(["--gpus", "8", "--steps", "20", "--warmup", "5", "--dry-launch"])
    assert out.returncode == 0, out.stderr
    d = json.loads(out.stdout.strip())
    cmd = d["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nnodes=1" in cmd and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and 1024 < int(cmd[cmd.index("--master-port") + 1]) < 65536
    i = cmd.index(BENCH)
    assert cmd[i + 1:] == ["--gpus", "8", "--steps", "20", "--warmup", "5"]          # the children get the same arguments, minus --dry-launch
    assert d["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and d["ranks"] == 8


def test_one_gpu_and_launched_ranks_do_not_self_launch():
    # under a launcher (WORLD_SIZE set) --dry-launch is refused instead of spawning again
    out = _run(["--gpus", "2", "--dry-launch"], env={"WORLD_SIZE": "2", "RANK": "0"})
    assert out.returncode != 0 and "--dry-launch needs" in out.stderr


def test_self_launch_relays_rank0_line_and_failures(tmp_path):
    """The launcher part of bench.py with `torch.distributed.run` replaced by a stand-in module (PYTHONPATH shadow): it must relay exactly
    one JSON line, pass a child's non-zero exit code on, and stop a hung run at --launch-timeout."""
    pkg = tmp_path / "torch" / "distributed"
    pkg.mkdir(parents=True)
    (tmp_path / "torch" / "__init__.py").write_text("")
    (pkg / "__init__.py").write_text("")
    (pkg / "run.py").write_text(
        "import os, sys, time\n"
        "mode = os.environ.get('FAKE_RUN_MODE', 'ok')\n"
        "print('noise from a rank', flush=True)\n"
        "if mode == 'ok': print('{\"metric\": \"m\", \"argv\": \"%s\"}' % ' '.join(sys.argv[1:]), flush=True)\n"
        "if mode == 'fail': sys.exit(7)\n"
        "if mode == 'hang': time.sleep(600)\n")
    env = {"PYTHONPATH": str(tmp_path)}
    ok = _run(["--gpus", "2", "--steps", "3"], env=dict(env, FAKE_RUN_MODE="ok"))
    assert ok.returncode == 0, ok.stderr
    lines = [ln for ln in ok.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and "--nproc-per-node=2" in json.loads(lines[0])["argv"] and "noise from a rank" in ok.stderr
    bad = _run(["--gpus", "2"], env=dict(env, FAKE_RUN_MODE="fail"))
    assert bad.returncode == 7 and "exited with 7" in bad.stderr
    hung = _run(["--gpus", "2", "--launch-timeout", "3"], env=dict(env, FAKE_RUN_MODE="hang"), timeout=60)
    assert hung.returncode == 124 and "was stopped" in hung.stderr

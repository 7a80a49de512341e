"""Copy the round-4 measurements worth keeping from gpurun_out/<tag>/ into profiles/r04_* and derive the figures DESIGN.md and
profiles/README.md quote (run here, after tools/r04_profile.sh ran on the GPU box):   python tools/r04_summarize.py <tag>"""
import csv
import json
import re
import shutil
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
src = ROOT / "gpurun_out" / (sys.argv[1] if len(sys.argv) > 1 else "r04p")
dst = ROOT / "profiles"
commit = subprocess.run(["git", "rev-parse", "--short=12", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip()
copies = {"kernel_stats.csv": "r04_wavefront_kernel_stats.csv", "kernel_stats_one_launch_set_in_flight.csv": "r04_wavefront_kernel_stats_one_launch_set_in_flight.csv",
          "kernel_stats_1m_one_launch_set_in_flight.csv": "r04_1m_kernel_stats_one_launch_set_in_flight.csv",
          "pmc_bunny_summary.txt": "r04_wavefront_pmc_summary.txt", "pmc_1m_summary.txt": "r04_1m_pmc_summary.txt"}
for a, b in copies.items():
    if (src / a).exists():
        shutil.copyfile(src / a, dst / b)
for a, b in (("pmc_bunny/traffic.json", "r04_traffic_bunny.json"), ("pmc_1m/traffic.json", "r04_traffic_1m.json")):
    if (src / a).exists():
        d = json.load(open(src / a))
        d["commit"] = commit
        d["kind"] = "profiled_offline"
        json.dump(d, open(dst / b, "w"), indent=1)


def counters(path):
    out, cur = {}, None
    for line in open(path):
        if line.startswith("k_"):
            cur = line.strip()
            out[cur] = {}
        elif cur and "=" in line:
            for m in re.finditer(r"(\w+)=([0-9.e+]+)", line):
                out[cur].setdefault(m.group(1), float(m.group(2)))
    return out


def stats(path):
    return {r["Name"]: (float(r["AverageNs"]), int(r["Calls"])) for r in csv.DictReader(open(path))}


lines = []
for tag, pm, ks in (("bunny (configs[1]), bench.py's timed mode: batches of 8 frames", "pmc_bunny_summary.txt", "kernel_stats_one_launch_set_in_flight.csv"),
                    ("1M triangles (configs[4] scene, 4 spp), batches of 8 frames", "pmc_1m_summary.txt", "kernel_stats_1m_one_launch_set_in_flight.csv")):
    if not (src / pm).exists() or not (src / ks).exists():
        continue
    c, k = counters(src / pm), stats(src / ks)
    lines.append(f"== {tag}: traversal launches, ONE launch set in flight (RT_LANES=1); PMC sums over all launches of the run, durations from rocprofv3 --kernel-trace --stats of the same command")
    for name, v in c.items():
        if not name.startswith("k_trace"):
            continue
        key = name.split("<")[1].split(">")[0].replace("(anonymous namespace)::", "")
        hit = next(((d, n) for nm, (d, n) in k.items() if key in nm.replace("(anonymous namespace)::", "")), None)
        if hit is None:
            continue
        dur, calls = hit
        n = v.get("dispatches", calls)
        clk = v.get("GRBM_GUI_ACTIVE", 0) / 8 / max(n, 1)          # shader cycles per launch
        acc = v.get("TCP_TOTAL_CACHE_ACCESSES_sum", 0) / max(n, 1)
        hbm = (2 * v.get("FETCH_SIZE", 0) + v.get("WRITE_SIZE", 0)) * 1024 / max(n, 1)
        valu = v.get("SQ_INSTS_VALU", 0) / max(n, 1)
        lines.append(f"{name}\n   launches {int(n)}  avg duration {dur / 1e3:9.1f} us   clock {clk / dur:5.2f} GHz   TCP cache accesses / clk / CU {acc / max(clk, 1) / 256:5.2f}"
                     f"   L1 hit {1 - v.get('TCP_TCC_READ_REQ_sum', 0) / max(v.get('TCP_TOTAL_CACHE_ACCESSES_sum', 1), 1):4.2f}"
                     f"   L2 hit {v.get('TCC_HIT_sum', 0) / max(v.get('TCC_HIT_sum', 0) + v.get('TCC_MISS_sum', 0), 1):4.2f}\n"
                     f"   fabric / HBM bytes per launch (2 x FETCH_SIZE + WRITE_SIZE) {hbm / 1e6:9.1f} MB = {hbm / dur:6.1f} GB/s = {hbm / dur / 8000:5.3f} of 8 TB/s"
                     f"   VALU wave-instructions {valu:.3g} = {valu * 2 / (1024 * max(clk, 1)):4.2f} of the issue slots (2 clk each, 1024 SIMDs)"
                     f"   lane utilisation {v.get('SQ_THREAD_CYCLES_VALU', 0) / max(v.get('SQ_INSTS_VALU', 1), 1) / 64:4.2f}"
                     f"   wave time waiting on memory {v.get('SQ_WAIT_ANY', 0) / max(v.get('SQ_WAVE_CYCLES', 1), 1):4.2f}")
open(dst / "r04_derived.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))

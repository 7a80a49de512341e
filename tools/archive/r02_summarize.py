"""Copy the round-2 measurements worth keeping from gpurun_out/<tag>/ into profiles/r02_* and derive the figures DESIGN.md and
profiles/README.md quote (run here, after tools/r02_profile.sh ran on the GPU box):   python tools/r02_summarize.py r02d"""
import json
import re
import shutil
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
src = ROOT / "gpurun_out" / (sys.argv[1] if len(sys.argv) > 1 else "r02d")
dst = ROOT / "profiles"
commit = subprocess.run(["git", "rev-parse", "--short=12", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip()
copies = {"kernel_stats.csv": "r02_wavefront_kernel_stats.csv", "kernel_stats_one_frame_in_flight.csv": "r02_wavefront_kernel_stats_one_frame_in_flight.csv",
          "kernel_stats_frame_by_frame_one_in_flight.csv": "r02_wavefront_kernel_stats_frame_by_frame_one_in_flight.csv",
          "kernel_stats_1m_one_frame_in_flight.csv": "r02_1m_kernel_stats_one_frame_in_flight.csv", "pmc_bunny_summary.txt": "r02_wavefront_pmc_summary.txt",
          "pmc_1m_summary.txt": "r02_1m_pmc_summary.txt", "gather_pmc.txt": "r02_gather_microbench_pmc.txt", "gather.txt": "r02_gather_microbench.txt",
          "bench.json": "r02_bench_line.json", "bench_1m.json": "r02_bench_line_1m.json"}
for a, b in copies.items():
    if (src / a).exists():
        shutil.copyfile(src / a, dst / b)
for a, b in (("traffic_bunny.json", "r02_traffic_bunny.json"), ("traffic_1m.json", "r02_traffic_1m.json")):
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
        elif cur and "=" in line and not line.lstrip().startswith("lane_util"):
            for m in re.finditer(r"(\w+)=([0-9.e+]+)", line):
                out[cur][m.group(1)] = float(m.group(2))
    return out


def stats(path):
    import csv
    return {r["Name"]: float(r["AverageNs"]) for r in csv.DictReader(open(path))}


lines = []
for tag, pm, ks, frames, tj in (("bunny (configs[1]), frame by frame", "pmc_bunny_summary.txt", "kernel_stats_frame_by_frame_one_in_flight.csv", 7, "traffic_bunny.json"),
                                ("1M triangles (configs[4] scene, 4 spp)", "pmc_1m_summary.txt", "kernel_stats_1m_one_frame_in_flight.csv", 5, "traffic_1m.json")):
    if not (src / pm).exists():
        continue
    c, k = counters(src / pm), stats(src / ks)
    tr = json.load(open(src / tj))["kernels"]
    lines.append(f"== {tag}: traversal launches, one frame in flight; PMC sums over {frames} frames, durations from rocprofv3 --kernel-trace --stats")
    for name, v in c.items():
        if not name.startswith("k_trace"):
            continue
        dur = next((d for n, d in k.items() if name.split("<")[1].split(">")[0].replace("(anonymous namespace)::", "") in n.replace("(anonymous namespace)::", "")), None)
        if dur is None:
            continue
        clk = v.get("GRBM_GUI_ACTIVE", 0) / 8 / frames            # shader cycles per launch
        acc = v.get("TCP_TOTAL_CACHE_ACCESSES_sum", 0) / frames
        hbm = tr.get(name, {}).get("hbm_bytes_per_frame_corrected", 0)
        valu = v.get("SQ_INSTS_VALU", 0) / frames
        lines.append(f"{name}\n   avg duration {dur / 1e3:9.1f} us   clock {clk / dur:5.2f} GHz   TCP cache accesses / clk / CU {acc / max(clk, 1) / 256:5.2f}"
                     f"   L1 hit {1 - v.get('TCP_TCC_READ_REQ_sum', 0) / max(v.get('TCP_TOTAL_CACHE_ACCESSES_sum', 1), 1):4.2f}"
                     f"   L2 hit {v.get('TCC_HIT_sum', 0) / max(v.get('TCC_HIT_sum', 0) + v.get('TCC_MISS_sum', 0), 1):4.2f}\n"
                     f"   fabric / HBM bytes per launch (2 x FETCH_SIZE + WRITE_SIZE) {hbm / 1e6:9.1f} MB = {hbm / dur:6.1f} GB/s = {hbm / dur / 8000:5.3f} of 8 TB/s"
                     f"   VALU wave-instructions {valu:.3g} = {valu * 2 / (1024 * max(clk, 1)):4.2f} of the issue slots (2 clk each, 1024 SIMDs)"
                     f"   lane utilisation {v.get('SQ_THREAD_CYCLES_VALU', 0) / max(v.get('SQ_INSTS_VALU', 1), 1) / 64:4.2f}"
                     f"   wave time waiting on memory {v.get('SQ_WAIT_ANY', 0) / max(v.get('SQ_WAVE_CYCLES', 1), 1):4.2f}")
open(dst / "r02_derived.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))

"""Per-kernel sums of the rocprofv3 --pmc passes under <dir>/p*/ (one counter set per pass), the number of dispatches with work, derived
figures, and <dir>/traffic.json (HBM bytes per kernel, gfx950 FETCH_SIZE correction, stamped with the sha256 of the kernel sources)."""
import collections
import csv
import glob
import hashlib
import json
import pathlib
import re
import sys

out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(lambda: collections.defaultdict(set))      # kernel -> counter -> dispatch ids
busy = collections.defaultdict(lambda: collections.defaultdict(float))    # kernel -> dispatch -> SQ_WAVES (to tell launches with work)


def short(name):
    m = re.search(r'(k_\w+)(<[^>]*>)?', name)
    return (m.group(1) + (m.group(2) or '')) if m else None


for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if not k:
            continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[k][r["Counter_Name"]].add(r["Dispatch_Id"])
        if r["Counter_Name"] == "SQ_WAVES":
            busy[k][r["Dispatch_Id"]] += float(r["Counter_Value"])
tr = {}
for k, d in sorted(agg.items()):
    g = d.get
    n = max((len(v) for v in disp[k].values()), default=0)
    print(k)
    print('   dispatches=%d  ' % n + '  '.join(f"{c}={v:.4g}" for c, v in sorted(d.items())))
    line = []
    if g("SQ_INSTS_VALU") and g("SQ_THREAD_CYCLES_VALU"):
        line.append(f"lane_util={g('SQ_THREAD_CYCLES_VALU') / g('SQ_INSTS_VALU') / 64:.2f}")
    if g("SQ_WAVE_CYCLES") and g("SQ_WAIT_ANY"):
        line.append(f"wait_frac={g('SQ_WAIT_ANY') / g('SQ_WAVE_CYCLES'):.2f} issue_frac={g('SQ_ACTIVE_INST_ANY', 0) / g('SQ_WAVE_CYCLES'):.2f}")
    acc_per_clk = None
    if g("TCP_TOTAL_CACHE_ACCESSES_sum"):
        line.append(f"L1hit={1 - g('TCP_TCC_READ_REQ_sum', 0) / g('TCP_TOTAL_CACHE_ACCESSES_sum'):.2f} L2hit={g('TCC_HIT_sum', 0) / max(g('TCC_HIT_sum', 0) + g('TCC_MISS_sum', 0), 1):.2f}")
        if g("GRBM_GUI_ACTIVE"):
            clk = g("GRBM_GUI_ACTIVE") / 8          # rocprofv3 sums the 8 XCDs
            acc_per_clk = g("TCP_TOTAL_CACHE_ACCESSES_sum") / clk / 256
            line.append(f"tcp_accesses_per_clk_per_cu={acc_per_clk:.3f}")
            if g("SQ_INSTS_VALU"):
                line.append(f"valu_issue_frac(2clk)={g('SQ_INSTS_VALU') * 2 / (1024 * clk):.2f}")
    print('   ' + '  '.join(line))
    if "FETCH_SIZE" in d or "WRITE_SIZE" in d:
        tr[k] = {"dispatches": n, "launches_with_rays": n, "fetch_kb_raw": d.get("FETCH_SIZE", 0), "write_kb": d.get("WRITE_SIZE", 0),
                 "hbm_bytes_corrected": (2 * d.get("FETCH_SIZE", 0) + d.get("WRITE_SIZE", 0)) * 1024,
                 "tcp_cache_accesses": d.get("TCP_TOTAL_CACHE_ACCESSES_sum"), "tcp_accesses_per_clk_per_cu": acc_per_clk}
h = hashlib.sha256()
for name in sorted(("rt_wave.hip", "rt_wave.hpp", "rt_api.hip", "rt_frame.hpp", "rt_device_shade.hpp", "rt_device_math.hpp", "rt_device_analytic.hpp")):   # = bench.py kernel_source_sha()
    f = pathlib.Path(__file__).resolve().parent.parent / "opengl-raytracing_amd" / "csrc" / name
    h.update(f.name.encode())
    h.update(f.read_bytes())
json.dump({"kernel_source_sha256": h.hexdigest(), "commit": None, "mode": "bench.py timed mode: rt_render_frames in batches of 8, RT_LANES=1 under the profiler",
           "note": "FETCH_SIZE / WRITE_SIZE in KB summed over all dispatches of the kernel; gfx950 FETCH_SIZE under-reports wide reads by 2x (MI355X_MICROARCH.md): corrected = 2 x fetch + write",
           "kernels": tr}, open(out + "/traffic.json", "w"), indent=1)

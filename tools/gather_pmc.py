"""Per access shape of tools/gather.hip: PMC counters of the separate rocprofv3 --pmc passes, next to the kernel duration.
usage: gather_pmc.py <dir with p*/ passes> [gather.txt]
Dispatch order of gather_bench: for each lane mask (64, 32, 16 active lanes) the 7 shapes below, two launches each
(the second one is what gather.txt times)."""
import collections
import csv
import glob
import sys

SHAPES = ["64B/lane 4x dwordx4 (2-wide node)", "128B/lane 8x dwordx4 (4-wide node)", "32B/lane 2x dwordx4", "16B/lane 1x dwordx4",
          "cooperative 16 lines/instr x4", "64B/lane, 16 distinct nodes/wave", "64B/lane, one node/wave"]
LOADS = [4, 8, 2, 1, 4, 4, 4]          # dwordx4 instructions per step
LANES = [64, 32, 16]
STEPS, WAVES, CUS = 400, 1280 * 4, 256

d = sys.argv[1]
rows = collections.defaultdict(dict)   # dispatch id -> counter -> value
dur = {}
for f in glob.glob(d + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k<" not in r["Kernel_Name"] and not r["Kernel_Name"].startswith("void k"):
            continue
        i = int(r["Dispatch_Id"])
        rows[i][r["Counter_Name"]] = rows[i].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        if r.get("End_Timestamp") and r.get("Start_Timestamp"):
            dur[i] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
ids = sorted(rows)
print(f"{len(ids)} dispatches with counters")
names = sorted({c for v in rows.values() for c in v})
for k, i in enumerate(ids):
    if k % 2 == 0:
        continue                         # first launch of each pair = warm-up
    g = k // 2
    if g >= len(SHAPES) * len(LANES):
        break
    lanes, shp = LANES[g // 7], g % 7
    v = rows[i]
    lane_loads = WAVES * STEPS * lanes * LOADS[shp]
    ns = dur.get(i, 0)
    cyc = v.get("GRBM_GUI_ACTIVE", 0) / 8.0      # summed over the 8 XCDs
    line = f"lanes {lanes:2d}  {SHAPES[shp]:36s} lane-loads {lane_loads:.3g}  dur {ns / 1e3:7.1f} us"
    if cyc:
        line += f"  clk {cyc / max(ns, 1):.2f} GHz  lane-loads/clk/CU {lane_loads / cyc / CUS:.2f}"
    for c in names:
        if c == "GRBM_GUI_ACTIVE" or c not in v:
            continue
        line += f"  {c}={v[c]:.4g}"
        if cyc and c.startswith(("TCP_TOTAL", "TCP_TAGRAM", "TCP_TCC_READ", "TA_FLAT")):
            line += f" ({v[c] / lane_loads:.2f}/lane-load, {v[c] / cyc / CUS:.2f}/clk/CU)"
    print(line)

"""Concurrency analysis of a rocprofv3 kernel trace: busy time (union of kernel intervals), time with 1 / 2 / 3+ kernels in flight, time per kernel
family alone vs overlapped, for the last 60 % of the trace (steady state)."""
import csv
import re
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
ev = []
for r in rows:
    m = re.search(r'(k_\w+)(<[^>]*>)?', r["Kernel_Name"])
    name = m.group(1) + ("<" + ("any" if "true" in (m.group(2) or "").split(",")[1:2][0] else "closest") + "," + m.group(2).split("::")[-1].split(",")[0] + ">" if m and m.group(2) and "k_trace" in m.group(1) else "") if m else r["Kernel_Name"][:30]
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name))
ev.sort()
t0, t1 = ev[0][0], max(e[1] for e in ev)
lo = t0 + (t1 - t0) * 4 // 10
ev = [e for e in ev if e[0] >= lo]
pts = []
for s, e, n in ev:
    pts.append((s, 1, n))
    pts.append((e, -1, n))
pts.sort()
active = defaultdict(int)
depth_time = defaultdict(int)
alone = defaultdict(int)
total = defaultdict(int)
trace_active_time = 0
prev = pts[0][0]
for t, d, n in pts:
    dt = t - prev
    if dt > 0:
        k = sum(active.values())
        depth_time[min(k, 4)] += dt
        names = [a for a, c in active.items() if c > 0]
        for a in names:
            total[a] += dt
            if k == 1:
                alone[a] += dt
        if any("k_trace" in a for a in names):
            trace_active_time += dt
    active[n] += d
    prev = t
span = pts[-1][0] - pts[0][0]
print("span %.2f ms; kernels in flight: " % (span / 1e6) + "  ".join("%d%s: %.1f%%" % (k, "+" if k == 4 else "", 100.0 * v / span) for k, v in sorted(depth_time.items())))
print("time with at least one traversal kernel in flight: %.1f%%" % (100.0 * trace_active_time / span))
for a in sorted(total, key=lambda x: -total[x]):
    print("  %-34s in flight %.1f%% of the span, alone %.1f%%" % (a, 100.0 * total[a] / span, 100.0 * alone[a] / span))

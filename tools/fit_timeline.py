#!/usr/bin/env python3
"""Per-launch timeline of the LAST fit in a rocprofv3 kernel trace (tools/fit_trace.py drive ...): start, duration, queue --
launches that overlap in time show as such (the double two-level fit runs on three streams).
    python tools/fit_timeline.py TRACE.csv [--all]     (default: potrf_step chains folded into one line each)"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "scale_x_kernel" in r["Kernel_Name"]][-1]
f = rows[idx:]
t0 = int(f[0]["Start_Timestamp"])
agg = []
for r in f:
    k = r["Kernel_Name"].split("(")[0].replace("void gpso::", "")
    s, e, q = int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "")
    if "--all" not in sys.argv and agg and agg[-1][0] == k and agg[-1][5] == q and "potrf_step" in k:
        agg[-1][2] = e
        agg[-1][3] += 1
    else:
        agg.append([k, s, e, 1, r.get("Grid_Size", ""), q])
end = max(a[2] for a in agg)
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in f)
print(f"span {(end - t0) / 1e3:.1f} us, sum of kernel durations {busy / 1e3:.1f} us, {len(f)} launches")
for k, s, e, c, g, q in agg:
    print(f"+{(s - t0) / 1e3:8.1f} .. {(e - t0) / 1e3:8.1f}  {(e - s) / 1e3:8.1f} us x{c:3d} q{q:>2} grid {g:>8} {k[:56]}")

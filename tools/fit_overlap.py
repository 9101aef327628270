#!/usr/bin/env python3
"""Print the launches of the last fit in a rocprofv3 kernel trace with start / end offsets (us), to see
which kernels overlap: python3 tools/fit_overlap.py <kernel_trace.csv> [first] [count]"""
import sys, csv
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "scale_x_kernel" in r["Kernel_Name"]]
f = rows[starts[-1]:]
t0 = int(f[0]["Start_Timestamp"])
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
count = int(sys.argv[3]) if len(sys.argv) > 3 else 80
for r in f[first:first + count]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    name = r["Kernel_Name"].split("(")[0].replace("void gpso::", "")[:40]
    print(f"{s / 1e3:10.1f} {e / 1e3:10.1f} {(e - s) / 1e3:8.1f}  q{r.get('Queue_Id', '?')} grid {r.get('Grid_Size', r.get('Grid_Size_X', '?'))}  {name}")

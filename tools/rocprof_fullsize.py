#!/usr/bin/env python3
"""Mean duration of the FULL-SIZE launches of one kernel in a rocprofv3 --kernel-trace CSV (the *_kernel_stats.csv of
the same run averages every launch of the name, the self-test's short ones included).

    rocprof_fullsize.py TRACE_DIR "KERNEL PATTERN" [OUT.json]
"""
import csv
import glob
import json
import sys

root, pat = sys.argv[1:3]
rows = []
for f in glob.glob(root + "/**/*kernel_trace.csv", recursive=True):
    with open(f) as fh:
        rows += [r for r in csv.DictReader(fh) if pat in r.get("Kernel_Name", "")]
if not rows:
    sys.exit(f"no launch of '{pat}' under {root}")
grid = lambda r: int(r.get("Grid_Size", 0) or 0) or int(r.get("Grid_Size_X", 0) or 0) * int(r.get("Grid_Size_Y", 1) or 1)
big = max(grid(r) for r in rows)
full = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows if grid(r) == big]
rec = {"kernel_pattern": pat, "launches": len(rows), "full_size_launches": len(full), "grid_threads": big,
       "mean_ns_full_size": sum(full) / len(full), "min_ns": min(full), "max_ns": max(full),
       "mean_ns_all_launches": sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows) / len(rows)}
print(json.dumps(rec))
if len(sys.argv) > 3:
    with open(sys.argv[3], "w") as fh:
        json.dump(rec, fh, indent=1)

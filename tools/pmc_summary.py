#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter CSVs: per-kernel mean of each counter per dispatch.  A kernel name
is launched with several grid sizes in one bench run (the headline batch, the self-test's N leaves, the
accuracy sample of the split-bf16 report): only the dispatches with the LARGEST grid of each kernel are
averaged -- a mean over all of them would dilute the per-launch figures."""
import csv, glob, sys, collections
root = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else "leaf_tiles"
rows = []
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            if pat in row.get("Kernel_Name", ""):
                rows.append(row)
biggest = collections.defaultdict(int)
for r in rows:
    k = r["Kernel_Name"].split("(")[0][-70:]
    biggest[k] = max(biggest[k], int(r.get("Grid_Size", 0) or 0))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r["Kernel_Name"].split("(")[0][-70:]
    if int(r.get("Grid_Size", 0) or 0) == biggest[k]:
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print(f"{k}   [grid {biggest[k]}]")
    for c, v in sorted(cs.items()):
        print(f"   {c:32s} mean {sum(v)/len(v):.6g}  (n={len(v)})")

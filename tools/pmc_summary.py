#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter CSVs: per-kernel mean of each counter (per dispatch)."""
import csv, glob, sys, collections
root = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else "leaf_tiles"
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            name = row.get("Kernel_Name", "")
            if pat in name:
                acc[name.split("(")[0][-60:]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        print(f"   {c:32s} mean {sum(v)/len(v):.6g}  (n={len(v)})")

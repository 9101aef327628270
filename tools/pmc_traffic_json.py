#!/usr/bin/env python3
"""rocprofv3 --pmc pass directories -> the traffic record bench.py reads (profiles/rNN_pmc_leaf_tiles_<math>_<workload>.json).

    pmc_traffic_json.py OUT.json WORKLOAD "KERNEL PATTERN" "DESCRIPTION" PASSDIR [PASSDIR ...]

Only the dispatches with the largest grid of the kernel are averaged (the self-test's and the accuracy sample's short
launches are left out).  gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE counts 64 B per 128-B request
for wide coalesced streaming reads -> doubled; WRITE_SIZE is exact.  Both counters are in KB.
"""
import collections
import csv
import glob
import json
import sys

out, workload, pat, desc = sys.argv[1:5]
rows = []
for root in sys.argv[5:]:
    for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
        with open(f) as fh:
            rows += [r for r in csv.DictReader(fh) if pat in r.get("Kernel_Name", "")]
if not rows:
    sys.exit(f"no dispatch of '{pat}' in {sys.argv[5:]}")
big = max(int(r.get("Grid_Size", 0) or 0) for r in rows)
acc = collections.defaultdict(list)
for r in rows:
    if int(r.get("Grid_Size", 0) or 0) == big:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
rec = {"workload": workload, "kernel": desc, "grid_threads": big,
       "source": "rocprofv3 --pmc, separate passes (tools/collect_profiles.sh), mean over the full-size dispatches only",
       "dispatches_averaged": {k: len(v) for k, v in acc.items()}}
for k, v in sorted(acc.items()):
    rec[k + ("_KB" if k in ("FETCH_SIZE", "WRITE_SIZE") else "")] = sum(v) / len(v)
rec["correction"] = ("gfx950: FETCH_SIZE counts 64 B per 128-B request for wide coalesced streaming reads (16 B/lane, global_load and "
                     "global_load ... lds alike) -> doubled; WRITE_SIZE exact (MI355X_MICROARCH.md, HBM section)")
if "FETCH_SIZE" in acc and "WRITE_SIZE" in acc:
    rec["traffic_bytes_per_launch"] = int(2 * rec["FETCH_SIZE_KB"] * 1024 + rec["WRITE_SIZE_KB"] * 1024)
if "SQ_VALU_MFMA_BUSY_CYCLES" in acc and "GRBM_GUI_ACTIVE" in acc:
    rec["matrix_pipe_busy"] = rec["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / (rec["GRBM_GUI_ACTIVE"] / 8.0)
with open(out, "w") as fh:
    json.dump(rec, fh, indent=1)
print(json.dumps(rec))

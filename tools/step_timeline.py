"""Timeline of a few bench steps from a rocprofv3 kernel trace: python tools/step_timeline.py <trace dir> [kernel substring]
(on the GPU box: rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline)."""
import csv
import glob
import sys

f = sorted(glob.glob(sys.argv[1] + "/*/*kernel_trace.csv"))[-1]
pat = sys.argv[2] if len(sys.argv) > 2 else "leaf_tiles"
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if pat in r["Kernel_Name"]]
i = idx[len(idx) // 3]  # (the tail of a bench run is its comparison of the other predict maths)
j = idx[len(idx) // 3 + 2]
t0 = int(rows[i - 1]["Start_Timestamp"])
prev_end = None
for r in rows[i - 1:j + 1]:
    n = r["Kernel_Name"].split("(")[0].replace("void gpso::", "")[:60]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = "" if prev_end is None else f"gap {(s - prev_end) / 1e3:6.1f}"
    print(f"{(s - t0) / 1e3:9.1f} -> {(e - t0) / 1e3:9.1f}  ({(e - s) / 1e3:7.1f} us) {gap:>12}  {n}")
    prev_end = e

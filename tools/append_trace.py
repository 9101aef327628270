#!/usr/bin/env python3
"""gpso_append under rocprofv3: which kernels an append runs, how long each takes and how long the device idles between them.

    rocprofv3 --kernel-trace --output-format csv -d OUT -- python3 tools/append_trace.py drive c3 [k] [dtype]
    python3 tools/append_trace.py analyse OUT      -> the LAST append of the run: one line per launch (start, duration, gap)
"""
import csv
import glob
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SHAPES = {"c3": (12, 2048), "c4": (20, 8192), "c5": (40, 16384)}


def drive(shape, k, dtype, reps=6):
    from pygpso_amd import HipGPEngine
    from tests.helpers import synthetic_leaves, synthetic_problem

    d, n = SHAPES[shape]
    X, y = synthetic_problem(n, d, seed=0)
    theta = ("Matern52", 0.25 * math.sqrt(d), 1.0, 1e-3, float(y.mean()))
    eng = HipGPEngine(dtype)
    for _ in range(reps):
        eng.set_data(X[:n - k], y[:n - k])
        eng.fit_eval(*theta, want_grad=False)
        eng.predict(synthetic_leaves(256, d))  # (the predict-ready pieces exist: the append repacks them in place)
        eng.synchronize()
        _, in_place = eng.append(X[n - k:], y[n - k:])
        print(f"append {eng.last_ms(2):.4f} ms in_place={in_place}")


def analyse(out):
    path = sorted(glob.glob(os.path.join(out, "**", "*kernel_trace.csv"), recursive=True))[0]
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    starts = [i for i, r in enumerate(rows) if "append_cross_kernel" in r["Kernel_Name"]]
    i0 = starts[-1]
    seq = []
    for r in rows[i0:]:
        name = r["Kernel_Name"].split("(")[0].replace("void gpso::", "")
        if seq and not ("append" in name or "pack_linv" in name):
            break
        seq.append((name, int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Grid_Size", "")))
    t0 = seq[0][1]
    last = t0
    busy = 0
    for name, s, e, grid in seq:
        print(f"  +{(s - t0) / 1e3:8.2f} us  {(e - s) / 1e3:8.2f} us  gap {(s - last) / 1e3:6.2f}  grid {grid:>9s}  {name[:70]}")
        busy += e - s
        last = e
    print(f"append: {len(seq)} launches, span {(last - t0) / 1e3:.2f} us, kernel time {busy / 1e3:.2f} us, gaps {(last - t0 - busy) / 1e3:.2f} us")


if __name__ == "__main__":
    if sys.argv[1] == "analyse":
        analyse(sys.argv[2])
    else:
        drive(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 7, sys.argv[4] if len(sys.argv) > 4 else "float32")

#!/usr/bin/env python3
"""Fit-only driver for rocprofv3 timelines: runs posterior fits (and NLML+grad evaluations) at one
size; `tools/fit_trace.py analyse <kernel_trace.csv>` then prints kernel time vs idle gaps per fit."""
import sys, os, csv
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def drive(n, d, dtype, reps, grad):
    from pygpso_amd import HipGPEngine
    from tests.helpers import synthetic_problem
    X, y = synthetic_problem(n, d, seed=0)
    eng = HipGPEngine(dtype)
    if os.environ.get("GPSO_FIT_SINGLE_MAX"):
        eng.set_fit_single_level_max(int(os.environ["GPSO_FIT_SINGLE_MAX"]))
    eng.set_data(X, y)
    ls = np.array([0.25 * np.sqrt(d)])
    for _ in range(reps):
        eng.fit_eval("Matern52", ls, 1.0, 1e-3, float(y.mean()), want_grad=grad)
        print(f"fit_ms(dev) {eng.last_ms(2):.3f}")


def analyse(path):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # split into fits at scale_x_kernel
    fits, cur = [], None
    for r in rows:
        if "scale_x_kernel" in r["Kernel_Name"]:
            cur = []
            fits.append(cur)
        if cur is not None:
            cur.append(r)
    f = fits[-1]
    t0, t1 = int(f[0]["Start_Timestamp"]), int(f[-1]["End_Timestamp"])
    busy, by = 0, {}
    last_end, gaps = t0, 0
    for r in f:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        busy += e - s
        k = r["Kernel_Name"].split("(")[0].replace("void gpso::", "")
        c = by.setdefault(k, [0, 0])
        c[0] += 1
        c[1] += e - s
        if s > last_end:
            gaps += s - last_end
        last_end = max(last_end, e)
    print(f"last fit: span {(t1 - t0) / 1e3:.1f} us, kernel time {busy / 1e3:.1f} us, idle gaps {gaps / 1e3:.1f} us, "
          f"{len(f)} launches")
    for k, (c, t) in sorted(by.items(), key=lambda kv: -kv[1][1]):
        print(f"  {k[:60]:60s} {c:5d} {t / 1e3:9.1f} us")


if __name__ == "__main__":
    if sys.argv[1] == "analyse":
        analyse(sys.argv[2])
    else:
        n, d = int(sys.argv[1]), int(sys.argv[2])
        dtype = sys.argv[3] if len(sys.argv) > 3 else "float32"
        drive(n, d, dtype, 6, len(sys.argv) > 4 and sys.argv[4] == "grad")

#!/usr/bin/env python3
"""Times gpso_append (k new points at fixed hyper-parameters) beside the from-scratch posterior fit it replaces, at the
BASELINE shapes.  python tools/append_bench.py [c3 c4 c5] [--k 1 7 64] [--dtype float32] -> one JSON line per case."""
import argparse
import json
import math
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SHAPES = {"c2": (6, 256), "c3": (12, 2048), "c4": (20, 8192), "c5": (40, 16384), "n300": (5, 300)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("shapes", nargs="*", default=["c3"])
    ap.add_argument("--k", type=int, nargs="*", default=[1, 7, 64])
    ap.add_argument("--dtype", default="float32")
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    from pygpso_amd import HipGPEngine
    from tests.helpers import synthetic_problem

    for name in args.shapes:
        d, n = SHAPES[name]
        X, y = synthetic_problem(n, d, seed=0)
        theta = ("Matern52", 0.25 * math.sqrt(d), 1.0, 1e-3, float(y.mean()))
        eng = HipGPEngine(args.dtype)
        for k in args.k:
            fit, app = [], []
            for _ in range(args.reps):
                eng.set_data(X[:n - k], y[:n - k])
                eng.fit_eval(*theta, want_grad=False)
                fit.append(eng.last_ms(2))
                _, in_place = eng.append(X[n - k:], y[n - k:])
                app.append(eng.last_ms(2) if in_place else float("nan"))
            print(json.dumps({"shape": name, "N": n, "D": d, "k": k, "dtype": args.dtype, "posterior_fit_ms": float(np.median(fit)),
                              "append_ms": float(np.median(app)), "ratio": float(np.median(fit) / np.median(app))}), flush=True)
        eng.close()


if __name__ == "__main__":
    main()

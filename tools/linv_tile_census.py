#!/usr/bin/env python3
"""Census of the L^-1 fragments the fp16-split predict kernel multiplies by (one MFMA A fragment = 16 rows x 32 columns):
max |entry| of every fragment of the lower triangle relative to the matrix maximum (the scale of the fp16 split), as a
histogram over powers of two.  A fragment 2^-11 or more below the scale has a SECOND fp16 piece below 2^-22 of it: its low
piece products (2 of the 3 MFMAs) could be skipped inside the error the dropped low x low product already costs -- IF
there are enough of them.  CPU only (the float64 oracle's factor): python tools/linv_tile_census.py [c3 c4] [--noise 1e-3]"""
import argparse
import json
import math
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SHAPES = {"c2": (6, 256), "c3": (12, 2048), "c4": (20, 8192), "c5": (40, 16384), "n4096d6": (6, 4096)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("shapes", nargs="*", default=["c3"])
    ap.add_argument("--noise", type=float, nargs="*", default=[1e-3])
    args = ap.parse_args()
    import scipy.linalg as sl

    from oracle import gpr
    from tests.helpers import synthetic_problem

    for name in args.shapes:
        d, n = SHAPES[name]
        X, y = synthetic_problem(n, d, seed=0)
        for noise in args.noise:
            th = gpr.Theta("Matern52", np.array([0.25 * math.sqrt(d)]), 1.0, noise, float(y.mean()))
            post = gpr.posterior(th, X, y)
            Li = sl.solve_triangular(post.L, np.eye(n), lower=True)
            gmax = float(np.max(np.abs(Li)))
            nr, nk = n // 16, n // 32
            fm = np.abs(Li).reshape(nr, 16, nk, 32).max(axis=(1, 3))  # [row tile][k-step]
            live = np.array([[32 * q <= 16 * r + 15 for q in range(nk)] for r in range(nr)])  # fragments that touch the lower triangle
            rel = fm[live] / gmax
            lg = np.floor(np.log2(np.maximum(rel, 1e-300))).astype(int)
            hist = {int(k): int(np.sum(lg == k)) for k in range(0, -25, -1) if np.sum(lg == k)}
            below = {f"<=2^-{b}": float(np.mean(rel <= 2.0 ** -b)) for b in (8, 9, 10, 11, 12, 14)}
            print(json.dumps({"shape": name, "N": n, "D": d, "noise": noise, "max_abs_linv": gmax, "fragments": int(rel.size),
                              "fraction_of_fragments_below": below, "log2_histogram": hist,
                              "median_log2": float(np.median(np.log2(np.maximum(rel, 1e-300))))}), flush=True)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Posterior fit and NLML+gradient times (library HIP events, median of 5) per fit-plane mode at the two-level shapes.
python tools/fit_bench.py [c4 c5 n4096] [--modes 2 1 0] -> one JSON line per case, with the float errors of the factor
against the float64 oracle where N <= 8192 (the oracle's CPU factorisation)."""
import argparse
import json
import math
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SHAPES = {"n4096": (6, 4096), "c4": (20, 8192), "c5": (40, 16384), "n4608": (12, 4608)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("shapes", nargs="*", default=["c4", "c5"])
    ap.add_argument("--modes", type=int, nargs="*", default=[2, 1])
    ap.add_argument("--noise", type=float, default=1e-3)
    ap.add_argument("--oracle", action="store_true")
    args = ap.parse_args()
    from pygpso_amd import HipGPEngine, _lib as L
    from tests.helpers import synthetic_problem

    for name in args.shapes:
        d, n = SHAPES[name]
        X, y = synthetic_problem(n, d, seed=0)
        theta = ("Matern52", 0.25 * math.sqrt(d), 1.0, args.noise, float(y.mean()))
        ref = None
        if args.oracle and n <= 8192:
            from oracle import gpr

            th = gpr.Theta(theta[0], np.array([theta[1]]), theta[2], theta[3], theta[4])
            ref = gpr.nlml_and_grad(th, X, y)
        for mode in args.modes:
            eng = HipGPEngine("float32")
            eng._check(eng._lib.gpso_set_option(eng._h, L.OPT_FIT_BF16_SYRK, mode))
            eng.set_data(X, y)
            out = {"shape": name, "N": n, "D": d, "fit_planes_mode": mode, "noise": args.noise}
            for key, grad in (("posterior_ms", False), ("nlml_grad_ms", True)):
                ts = []
                for _ in range(5):
                    f, g = eng.fit_eval(*theta, want_grad=grad)
                    ts.append(eng.last_ms(2))
                out[key] = float(np.median(ts))
            out["nlml"] = f
            if ref is not None:
                out["nlml_rel_err"] = abs(f - ref[0]) / abs(ref[0])
                out["grad_err"] = float(np.max(np.abs(g - ref[1]) / np.maximum(1.0, np.abs(ref[1]))))
            print(json.dumps(out), flush=True)
            eng.close()


if __name__ == "__main__":
    main()

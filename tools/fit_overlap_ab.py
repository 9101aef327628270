#!/usr/bin/env python3
"""The double two-level fit with and without the overlapped schedule (GPSO_OPT_FIT_OVERLAP), alternating in one process:
device ms of the posterior fit and of an NLML + gradient evaluation, and whether the results are the same bits.

    python tools/fit_overlap_ab.py [n4096 c4 c5] [--dtype float64] -> one JSON line per shape
"""
import argparse
import hashlib
import json
import math
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SHAPES = {"n4096": (6, 4096), "c4": (20, 8192), "c5": (40, 16384), "n3000": (5, 3000), "n6000": (12, 6000)}

ap = argparse.ArgumentParser()
ap.add_argument("shapes", nargs="*", default=["n4096", "c4"])
ap.add_argument("--dtype", default="float64")
ap.add_argument("--reps", type=int, default=4)
ap.add_argument("--modes", type=int, nargs="*", default=[0, 3])
args = ap.parse_args()
from pygpso_amd import HipGPEngine, _lib as L  # noqa: E402
from tests.helpers import synthetic_leaves, synthetic_problem  # noqa: E402

for name in args.shapes:
    d, n = SHAPES[name]
    X, y = synthetic_problem(n, d, seed=0)
    theta = ("Matern52", 0.25 * math.sqrt(d), 1.0, 1e-3, float(y.mean()))
    Xs = synthetic_leaves(2048, d)
    engs = {}
    for ov in args.modes:
        e = HipGPEngine(args.dtype)
        e._check(e._lib.gpso_set_option(e._h, L.OPT_FIT_OVERLAP, ov))
        e.set_data(X, y)
        engs[ov] = e
    ms = {(ov, g): [] for ov in args.modes for g in (False, True)}
    sig = {}
    for rep in range(args.reps):
        for ov in args.modes:
            for grad in (False, True):
                f, g = engs[ov].fit_eval(*theta, want_grad=grad)
                ms[(ov, grad)].append(engs[ov].last_ms(2))
                if rep == args.reps - 1 and grad:
                    mean, var = engs[ov].predict(Xs)
                    sig[ov] = hashlib.sha1(np.float64(f).tobytes() + g.tobytes() + mean.tobytes() + var.tobytes()
                                           + engs[ov].get_vector(L.VEC_ALPHA).tobytes()).hexdigest()[:12]
    out = {"shape": name, "N": n, "D": d, "dtype": args.dtype,
           "posterior_ms": {str(ov): round(float(np.median(ms[(ov, False)])), 3) for ov in args.modes},
           "nlml_grad_ms": {str(ov): round(float(np.median(ms[(ov, True)])), 3) for ov in args.modes},
           "same_bits": len(set(sig.values())) == 1}
    print(json.dumps(out), flush=True)
    for e in engs.values():
        e.close()

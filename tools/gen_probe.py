#!/usr/bin/env python3
"""Float vs double generation of the cross-Gram tile: self-test readings and actual leaf errors against the
float64 oracle, per engine dtype, on the bench posterior (C3 at noise 1e-3 / 1e-6) and a C2-size one."""
import json, math, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import gpr
from pygpso_amd import HipGPEngine
from tests.helpers import synthetic_leaves, synthetic_problem

for n, d, noise in ((2048, 12, 1e-3), (2048, 12, 1e-6), (256, 6, 1e-3), (256, 6, 1e-6)):
    X, y = synthetic_problem(n, d, seed=0)
    Xs = synthetic_leaves(2048, d, seed=1)
    th = gpr.Theta("Matern52", 0.25 * math.sqrt(d), 1.0, noise, float(y.mean()))
    post = gpr.posterior(th, X, y)
    mref, vref = gpr.predict_y(post, Xs)
    for dtype in ("float32", "mixed"):
        for gen in ("float32", "float64", "auto"):
            eng = HipGPEngine(dtype, generation=gen, precision_check=True)
            eng.set_data(X, y)
            try:
                eng.fit_eval(th.kernel, th.lengthscales, th.variance, th.noise, th.mean_c, want_grad=False)
                info = eng.precision_info()
                eng.set_precision_check(False)
                m, v = eng.predict(Xs)
                rec = dict(n=n, d=d, noise=noise, dtype=dtype, gen=gen, chosen=info["generation"], passed=info["passed"],
                           st_var=info["max_abs_err_var"], st_mean=info["max_abs_err_mean"], tol_var=info["tol_var_abs"],
                           leaf_var=float(np.max(np.abs(v - vref))), leaf_mean=float(np.max(np.abs(m - mref))))
            except Exception as e:  # noqa: BLE001
                rec = dict(n=n, d=d, noise=noise, dtype=dtype, gen=gen, error=str(e)[:100])
            print(json.dumps(rec))

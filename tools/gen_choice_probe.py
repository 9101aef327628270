#!/usr/bin/env python3
"""Which generation arithmetic does GPSO_GEN_AUTO settle on, per shape, with the x.x* contraction on the fp16 pipe and on
the f32 instruction?  Prints the self-test's readings (errors at the training inputs, relative to their tolerances) and
the kernel time that follows from the choice."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

from pygpso_amd import HipGPEngine  # noqa: E402
from tests.helpers import synthetic_leaves, synthetic_problem  # noqa: E402

shapes = [(256, 6), (1024, 6), (2048, 6), (2048, 12), (4096, 6), (8192, 6), (2048, 3), (2048, 2), (512, 4), (3072, 8)]
for n, d in shapes:
    X, y = synthetic_problem(n, d, seed=0)
    Xs = synthetic_leaves(16384, d, seed=1).astype(np.float32)
    for which in ("auto", "f32"):
        eng = HipGPEngine("float32")
        eng.set_contraction(which)
        eng.set_data(X, y)
        eng.fit_eval("Matern52", np.array([0.25 * np.sqrt(d)]), 1.0, 1e-3 if n <= 4096 else 1e-2, float(y.mean()), want_grad=False)
        info = eng.precision_info()
        for _ in range(3):
            eng.best_ucb(Xs, 1.8)
        print(json.dumps({"N": n, "D": d, "contraction": which, "generation": info["generation"], "math": info["predict_math"],
                          "err_var_over_tol": info["max_abs_err_var"] / info["tol_var_abs"],
                          "err_mean_over_tol": info["max_abs_err_mean"] / info["tol_mean_abs"], "kernel_ms": eng.last_ms(0)}), flush=True)
        eng.close()

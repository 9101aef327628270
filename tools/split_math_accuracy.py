#!/usr/bin/env python3
"""
Error of every predict math (native f32 MFMA, bf16x6, f16x3, bf16x3) against the float64 oracle, on the posteriors of
tests/test_gpu_precision.py (noise at GPflow's floor, "mixed" engine: float64 factor, float apply) and on bench-like
posteriors (noise 1e-3, "float32" engine).  One line of JSON per (problem, math): max |d var| / sigma^2, max |d mean|,
what the self-test measured.  GPU only; the oracle is the checker.
"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import gpr  # noqa: E402
from pygpso_amd import HipGPEngine  # noqa: E402
from pygpso_amd._lib import GpsoPrecisionError  # noqa: E402
from tests.helpers import synthetic_leaves, synthetic_problem  # noqa: E402
from tests.test_gpu_precision import PROBLEMS, _problem  # noqa: E402

MODES = ("native", "bf16x6", "f16x3", "bf16x3")


def run(name, dtype, X, y, th, leaves, mean_ref, var_ref):
    for math in MODES:
        eng = HipGPEngine(dtype, predict_math=math, generation="float64", tol_var=1e-2, tol_mean=1e-2)  # (generation fixed: the apply is what differs)
        eng.set_data(X, y)
        eng.fit_eval(th.kernel, th.lengthscales, th.variance, th.noise, th.mean_c, want_grad=False)
        rec = {"problem": name, "engine": dtype, "math": math}
        try:
            mean, var = eng.predict(leaves)
            rec["dvar_over_sigma2"] = float(np.max(np.abs(var - var_ref)) / th.variance)
            rec["dmean"] = float(np.max(np.abs(mean - mean_ref)))
            rec["generation"] = eng.precision_info().get("generation")
        except GpsoPrecisionError as exc:
            rec["refused"] = str(exc)[:120]
        print(json.dumps(rec), flush=True)
        eng.close()


def main():
    for name in sorted(PROBLEMS):
        X, y, th, leaves, post, mean_ref, var_ref = _problem(name)
        run(name, "mixed", X, y, th, leaves, mean_ref, var_ref)
    for n, d, kern in ((2048, 12, "Matern52"), (1024, 40, "Matern32"), (512, 12, "SquaredExponential"), (4096, 20, "Matern52")):
        X, y = synthetic_problem(n, d, seed=1)
        th = gpr.Theta(kern, 0.25 * np.sqrt(d) * np.ones(1), 1.0, 1e-3, float(y.mean()))
        leaves = synthetic_leaves(2048, d)
        post = gpr.posterior(th, X, y)
        mean_ref, var_ref = gpr.predict_y(post, leaves)
        for dtype in ("float32", "mixed"):
            run(f"N{n}-D{d}-{kern}-noise1e-3", dtype, X, y, th, leaves, mean_ref, var_ref)


if __name__ == "__main__":
    main()

import json, os, sys
sys.path.insert(0, "/root/repo")
import numpy as np
from pygpso_amd import HipGPEngine
from tests.helpers import synthetic_leaves, synthetic_problem
for n, d in [(2048, 6), (2048, 3), (512, 4), (2048, 12), (1024, 6)]:
    X, y = synthetic_problem(n, d, seed=0)
    for which in ("auto", "f32"):
        for gen in ("float32", "float64"):
            eng = HipGPEngine("float32", generation=gen)
            eng.set_contraction(which)
            eng.set_data(X, y)
            eng.fit_eval("Matern52", np.array([0.25 * np.sqrt(d)]), 1.0, 1e-3, float(y.mean()), want_grad=False)
            info = eng.precision_info()
            print(json.dumps({"N": n, "D": d, "contraction": which, "gen": gen, "generation": info["generation"], "err_var_over_tol": round(info["max_abs_err_var"] / info["tol_var_abs"], 5),
                              "err_mean_over_tol": round(info["max_abs_err_mean"] / info["tol_mean_abs"], 5), "passed": info["passed"]}), flush=True)
            eng.close()

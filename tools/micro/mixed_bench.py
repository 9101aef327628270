import sys, json
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from pygpso_amd import HipGPEngine
from tests.helpers import synthetic_leaves, synthetic_problem
n, d, m = 2048, 12, 65536
X, y = synthetic_problem(n, d, seed=0)
leaves = torch.from_numpy(synthetic_leaves(m, d).astype(np.float32)).cuda()
for dtype, noise, gen in (("float32", 1e-3, None), ("mixed", 1e-3, None), ("mixed", 1e-6, None), ("mixed", 1e-6, "float64"), ("mixed", 1e-3, "float32")):
    eng = HipGPEngine(dtype, **({} if gen is None else {"generation": gen}))
    eng.set_data(X, y)
    eng.fit_eval("Matern52", np.array([0.25 * np.sqrt(d)]), 1.0, noise, float(y.mean()), want_grad=False)
    info = eng.precision_info()
    ks = []
    for _ in range(30):
        eng.best_ucb(leaves, 2.0)
        ks.append(eng.last_ms(0))
    print(json.dumps({"dtype": dtype, "noise": noise, "gen_opt": gen, "generation": info["generation"], "math": info["predict_math"], "passed": info["passed"], "kernel_ms": float(np.median(ks[5:]))}), flush=True)
    eng.close()

import sys, json, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from pygpso_amd import HipGPEngine
from tests.helpers import synthetic_leaves, synthetic_problem
for n in ([int(v) for v in sys.argv[1:]] or [300, 600, 1100, 2100]):
    d, m = 8, 65536
    X, y = synthetic_problem(n, d, seed=0)
    leaves = torch.from_numpy(synthetic_leaves(m, d).astype(np.float32)).cuda()
    eng = HipGPEngine("float32")
    eng.set_data(X, y)
    ts = []
    for _ in range(8):
        eng.fit_eval("Matern52", np.array([0.25 * np.sqrt(d)]), 1.0, 1e-3, float(y.mean()), want_grad=False)
        ts.append(eng.last_ms(2))
    ks = []
    for _ in range(12):
        eng.best_ucb(leaves, 2.0)
        ks.append(eng.last_ms(0))
    print(json.dumps({"N": n, "npad": eng.padded_n, "math": eng.precision_info()["predict_math"], "fit_ms": float(np.median(ts[2:])), "kernel_ms": float(np.median(ks[2:]))}), flush=True)
    eng.close()

import sys, json, hashlib
sys.path.insert(0, "/root/repo")
import numpy as np
from pygpso_amd import HipGPEngine
from tests.helpers import synthetic_problem
out = {}
for dtype in ("float32", "float64"):
    for n, d, ard in ((300, 5, False), (2048, 12, False), (1024, 20, True), (4096, 40, False)):
        X, y = synthetic_problem(n, d, seed=0)
        eng = HipGPEngine(dtype)
        eng.set_data(X, y)
        ls = (0.25 * np.sqrt(d) * (1 + 0.1 * np.arange(d))) if ard else np.array([0.25 * np.sqrt(d)])
        f, g = eng.fit_eval("Matern52", ls, 1.0, 1e-3, float(y.mean()), want_grad=True)
        out[f"{dtype} {n} {d} ard={ard}"] = hashlib.sha1(np.asarray([f]).tobytes() + np.asarray(g).tobytes()).hexdigest()[:12]
        eng.close()
print("GRADBITS " + json.dumps(out))

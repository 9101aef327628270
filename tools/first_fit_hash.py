"""Hash of the FIRST fit + predict of a fresh process (python tools/first_fit_hash.py N D [dtype] [math]): run it in a
loop and count the distinct hashes -- the packed-mean failure (profiles/r02h_packed_mean_bug.txt) showed on first
launches, which a probe that repeats inside one process sees only once."""
import sys, hashlib, numpy as np
sys.path.insert(0, ".")
from pygpso_amd import HipGPEngine, _lib as L
from tests.helpers import synthetic_problem, synthetic_leaves
n, d = int(sys.argv[1]), int(sys.argv[2])
X, y = synthetic_problem(n, d, seed=5)
dtype = sys.argv[3] if len(sys.argv) > 3 else "float32"
math = sys.argv[4] if len(sys.argv) > 4 else "auto"
e = HipGPEngine(dtype, predict_math=math); e.set_data(X, y)
f, g = e.fit_eval("Matern52", 0.25 * np.sqrt(d) * np.ones(1), 1.0, 1e-2, float(y.mean()), want_grad=True)
mean, var = e.predict(synthetic_leaves(513, d, seed=9))
h = hashlib.sha1(np.float64(f).tobytes() + np.asarray(g).tobytes() + e.get_matrix(L.MAT_LINV).tobytes() + mean.tobytes() + var.tobytes()).hexdigest()[:12]
print(h)

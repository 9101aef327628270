import sys, math, time, cProfile, pstats, numpy as np, torch
sys.path.insert(0, '.')
from pygpso_amd import HipGPEngine
from tests.helpers import synthetic_problem, synthetic_leaves
X, y = synthetic_problem(2048, 12, seed=0)
lv = torch.from_numpy(synthetic_leaves(65536, 12, seed=1).astype(np.float32)).cuda()
eng = HipGPEngine("float32"); eng.set_timing(False)
eng.set_data(X, y); eng.fit_eval("Matern52", 0.25 * math.sqrt(12), 1.0, 1e-3, float(y.mean()), want_grad=False)
for _ in range(200): eng.best_ucb(lv, 1.82)
pr = cProfile.Profile(); pr.enable()
for _ in range(1000): eng.best_ucb(lv, 1.82)
pr.disable()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(12)

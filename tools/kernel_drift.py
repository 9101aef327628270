import sys, time, math, numpy as np, torch
sys.path.insert(0, '.')
from pygpso_amd import HipGPEngine
from tests.helpers import synthetic_problem, synthetic_leaves
X, y = synthetic_problem(2048, 12, seed=0)
lv = torch.from_numpy(synthetic_leaves(65536, 12, seed=1).astype(np.float32)).cuda()
eng = HipGPEngine("float32")
eng.set_data(X, y); eng.fit_eval("Matern52", 0.25 * math.sqrt(12), 1.0, 1e-3, float(y.mean()), want_grad=False)
ks = []
t0 = time.perf_counter()
for i in range(1200):
    eng.best_ucb(lv, 1.82); ks.append((time.perf_counter() - t0, eng.last_ms(0)))
ks = np.array(ks)
for lo in range(0, 1200, 100):
    seg = ks[lo:lo + 100, 1]
    print(f"steps {lo:4d}-{lo+99:4d}  t={ks[lo,0]:.3f}s  kernel ms: mean {seg.mean():.4f} min {seg.min():.4f} max {seg.max():.4f}")

#!/usr/bin/env python3
"""Host-side cost of one NLML+gradient evaluation at small N (run on the GPU box): wall time per
engine.fit_eval / HipGPR.training_loss call vs the device time the library reports, plus a cProfile
of a full surrogate update."""
import sys, os, time, cProfile, pstats
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygpso_amd import HipGPEngine
from tests.helpers import synthetic_problem

n, d = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (80, 2)
X, y = synthetic_problem(n, d, seed=0)
eng = HipGPEngine("float64")
eng.set_data(X, y)
ls = np.array([0.25 * np.sqrt(d)])
for _ in range(20):
    eng.fit_eval("Matern52", ls, 1.0, 1e-3, float(y.mean()), want_grad=True)
reps = 500
t = time.perf_counter()
dev = 0.0
for _ in range(reps):
    eng.fit_eval("Matern52", ls, 1.0, 1e-3, float(y.mean()), want_grad=True)
    dev += eng.last_ms(2)
wall = (time.perf_counter() - t) / reps * 1e3
print(f"engine.fit_eval n={n}: wall {wall:.3f} ms/call, device {dev / reps:.3f} ms/call")
# the optimiser's own call: unconstrained variables in, (f, df/du) out -- transforms and chain rule inside the library
from oracle import gpr  # (only for the inverse softplus of the starting point)
u = np.array([gpr.softplus_inv(ls[0]), gpr.softplus_inv(1.0), gpr.softplus_inv(1e-3 - 1e-6), float(y.mean())])
for _ in range(20):
    eng.fit_eval_u("Matern52", u, 1, True)
t = time.perf_counter()
dev = 0.0
for _ in range(reps):
    eng.fit_eval_u("Matern52", u, 1, True)
    dev += eng.last_ms(2)
wall = (time.perf_counter() - t) / reps * 1e3
print(f"engine.fit_eval_u n={n}: wall {wall:.3f} ms/call, device {dev / reps:.3f} ms/call")
eng.set_timing(False)  # GPSO_OPT_TIMING 0: what the drop-in surrogate's engine runs with
for _ in range(20):
    eng.fit_eval_u("Matern52", u, 1, True)
t = time.perf_counter()
for _ in range(reps):
    eng.fit_eval_u("Matern52", u, 1, True)
wall = (time.perf_counter() - t) / reps * 1e3
print(f"engine.fit_eval_u n={n}, timing events off: wall {wall:.3f} ms/call")
eng.set_timing(True)

from pygpso_amd import GPRSurrogate, kernels
surr = GPRSurrogate.default()
surr.append(X, y)
surr.gp_update()
t = time.perf_counter()
pr = cProfile.Profile()
pr.enable()
surr.gp_update()
pr.disable()
print(f"gp_update (L-BFGS-B fit + predict on the points): {(time.perf_counter() - t) * 1e3:.1f} ms")
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)

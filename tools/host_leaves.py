#!/usr/bin/env python3
"""best_ucb with host-resident leaves (the reference's calling convention: numpy float64 coords) vs
device-resident leaves, wall time per call at one size (run on the GPU box)."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pygpso_amd import HipGPEngine
from tests.helpers import synthetic_leaves, synthetic_problem

n, d, m = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (2048, 12, 65536)))
X, y = synthetic_problem(n, d, seed=0)
eng = HipGPEngine("float32")
eng.set_data(X, y)
eng.fit_eval("Matern52", np.array([0.25 * np.sqrt(d)]), 1.0, 1e-3, float(y.mean()), want_grad=False)
Xs64 = synthetic_leaves(m, d, seed=1)
Xs32 = Xs64.astype(np.float32)
dev32 = torch.from_numpy(Xs32).cuda()
vs = 1.8213863677184496


def bench(xs, label):
    for _ in range(3):
        eng.best_ucb(xs, vs)
    t = time.perf_counter()
    reps = 10
    for _ in range(reps):
        eng.best_ucb(xs, vs)
    print(f"{label:28s} {(time.perf_counter() - t) / reps * 1e3:8.3f} ms/call   (kernel {eng.last_ms(0):.3f} ms)")


bench(dev32, "device float32 (resident)")
bench(Xs32, "host float32 (numpy)")
bench(Xs64, "host float64 (numpy)")

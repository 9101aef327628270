#!/usr/bin/env python3
"""Accuracy + timing of the split-bf16 predict modes vs native f32 and the float64 oracle (GPU box)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import gpr
from pygpso_amd import HipGPEngine
from tests.helpers import synthetic_problem, synthetic_leaves
for (n, d, m, noise) in ((2048, 12, 65536, 1e-3), (512, 6, 4096, 1e-3), (1024, 40, 8192, 1e-2)):
    X, y = synthetic_problem(n, d, seed=0); Xs = synthetic_leaves(m, d)
    th = gpr.Theta("Matern52", 0.25 * np.sqrt(d), 1.0, noise, float(y.mean()))
    post = gpr.posterior(th, X, y)
    sub = np.random.default_rng(3).choice(m, min(m, 2048), replace=False)
    mref, vref = gpr.predict_y(post, Xs[sub])
    for mode in ("native", "bf16x6", "bf16x3"):
        eng = HipGPEngine("float32", predict_math=mode)
        eng.set_data(X, y); eng.fit_eval(th.kernel, th.lengthscales, th.variance, th.noise, th.mean_c, want_grad=False)
        mean, var = eng.predict(Xs.astype(np.float32))
        ts = []
        for _ in range(5):
            eng.best_ucb(Xs.astype(np.float32), gpr.VARSIGMA_DEFAULT); ts.append(eng.last_ms(0))
        print(f"n={n} d={d} m={m} {mode:7s}: mean err {np.abs(mean[sub]-mref).max():.2e} var err {np.abs(var[sub]-vref).max():.2e} "
              f"kernel {np.median(ts):.3f} ms")

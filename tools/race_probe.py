#!/usr/bin/env python3
"""Run-to-run determinism probe (run on the GPU box).  Every kernel of the path is deterministic by
construction (no atomics in the sums), so the SAME posterior and leaves must give the SAME bits every time:
a result that changes between runs is a race or a hazard (the packed accumulation of the means in the split-bf16
predict kernel, profiles/r02h_packed_mean_bug.txt, showed up this way: fuzz seed 22, D = 3).  Small D and short
workgroups are the sensitive corner.

  python tools/race_probe.py [REPEATS]
"""
import os, sys, itertools
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygpso_amd import HipGPEngine
from pygpso_amd._lib import GpsoPrecisionError
from tests.helpers import synthetic_problem, synthetic_leaves

REPEATS = int(sys.argv[1]) if len(sys.argv) > 1 else 60
total_bad = 0
cases = list(itertools.product(["float32", "mixed", "float64"], ["native", "f16x3", "bf16x6", "bf16x3"],
                               [1, 3, 6, 12, 40], [200, 512, 1024], ["Matern32", "SquaredExponential"]))
for dtype, math, d, n, kernel in cases:
    if dtype == "float64" and math != "native":
        continue
    X, y = synthetic_problem(n, d, seed=7 * n + d)
    Xs = synthetic_leaves(257, d, seed=11 * n + d)
    ls = 0.25 * np.sqrt(d) * np.ones(1)
    ref, nbad, refused = None, 0, False
    for _ in range(REPEATS):
        eng = HipGPEngine(dtype, predict_math=math)
        eng.set_data(X, y)
        try:
            f, g = eng.fit_eval(kernel, ls, 1.3, 1e-3, float(y.mean()), want_grad=True)
            mean, var = eng.predict(Xs)
        except GpsoPrecisionError:
            refused = True
            break
        cur = (np.float64(f).tobytes(), np.asarray(g).tobytes(), mean.tobytes(), var.tobytes())
        if ref is None:
            ref = cur
        nbad += cur != ref
    total_bad += nbad
    tag = "refused" if refused else ("ok " if nbad == 0 else "BAD")
    print(f"{tag} {dtype:7s} {math:7s} {kernel:18s} n={n:5d} d={d:2d}: {nbad} of {REPEATS} runs differ from the first")
print(f"{len(cases)} configurations, {total_bad} differing runs")
sys.exit(1 if total_bad else 0)

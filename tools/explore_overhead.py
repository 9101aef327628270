#!/usr/bin/env python3
"""Wall vs device time of one exploration level (gpso_best_ucb_grow on two boxes) at the reference's sizes."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import gpr, tree
from pygpso_amd import HipGPEngine
from tests.helpers import synthetic_problem

for n, d, depth in ((50, 2, 5), (80, 4, 7), (60, 6, 9)):
    X, y = synthetic_problem(n, d, seed=0)
    eng = HipGPEngine("float64")
    eng.set_data(X, y)
    eng.fit_eval("Matern52", [0.25 * np.sqrt(d)], 1.0, 1e-3, float(y.mean()), want_grad=False)
    kids = tree.split_bounds([(0.0, 1.0)] * d)
    boxes = np.array([kids[0], kids[2]])
    for _ in range(20):
        eng.best_ucb_grow(boxes, depth, gpr.VARSIGMA_DEFAULT)
    reps, dev, tile = 300, 0.0, 0.0
    t = time.perf_counter()
    for _ in range(reps):
        eng.best_ucb_grow(boxes, depth, gpr.VARSIGMA_DEFAULT)
        dev += eng.last_ms(1)
        tile += eng.last_ms(0)
    wall = (time.perf_counter() - t) / reps * 1e3
    print(f"best_ucb_grow n={n} d={d} depth={depth} rows scored {eng.last_count(0)} of {eng.last_count(1)}: "
          f"wall {wall:.3f} ms/call, device {dev / reps:.3f} ms/call, leaf-tile kernel {tile / reps:.3f} ms")

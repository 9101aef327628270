import sys, time, numpy as np
sys.path.insert(0, '.')
from pygpso_amd import HipGPEngine
from tests.helpers import synthetic_problem
from oracle import tree
for n, d, depth in [(52, 2, 5), (100, 4, 7), (128, 6, 9), (200, 3, 8)]:
    X, y = synthetic_problem(n, d, seed=0)
    kids = tree.split_bounds([(0.0, 1.0)] * d)
    boxes = np.array([kids[0], kids[2]])
    out = {}
    for mode in (1, 2, 3):
        eng = HipGPEngine("float64"); eng.set_timing(False); eng.set_small_calls(mode)
        eng.set_data(X, y); eng.fit_eval("Matern52", [0.3], 1.0, 1e-3, float(y.mean()), want_grad=False)
        for _ in range(300): r = eng.best_ucb_grow(boxes, depth, 1.82)
        t0 = time.perf_counter()
        for _ in range(2000): r = eng.best_ucb_grow(boxes, depth, 1.82)
        out[mode] = round((time.perf_counter() - t0) / 2000 * 1e6, 2)
    print(f"N={n} D={d} depth={depth} rows/box={(3**depth-1)//2}: wall us per best_ucb_grow  default(1) {out[1]} | three launches(2) {out[2]} | one launch(3) {out[3]}")

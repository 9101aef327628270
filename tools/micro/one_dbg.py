"""One-launch vs three-launch vs general sequence of a small best_ucb_grow call: wall us per call (same process, same box)."""
import sys, os, time, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from pygpso_amd import HipGPEngine
from oracle import tree
from tests.helpers import synthetic_problem
for n, d, depth in ((52, 2, 5), (100, 4, 7), (128, 6, 9), (256, 6, 8)):
    X, y = synthetic_problem(n, d, seed=0)
    kids = tree.split_bounds([(0.0, 1.0)] * d)
    boxes = np.array([kids[0], kids[2]])
    out = {}
    for mode in (1, 2, 0):
        eng = HipGPEngine("float64")
        eng.set_timing(False)
        eng.set_small_calls(mode)
        eng.set_data(X, y)
        eng.fit_eval("Matern52", [0.5], 1.0, 1e-3, 0.0, want_grad=False)
        for _ in range(30):
            r = eng.best_ucb_grow(boxes, depth, 1.8)
        t0 = time.perf_counter()
        for _ in range(300):
            r = eng.best_ucb_grow(boxes, depth, 1.8)
        out[mode] = ((time.perf_counter() - t0) / 300 * 1e6, int(r[0][0]), float(r[3][0]))
    print(f"N={n} D={d} depth={depth} rows/box={tree.grow_count(depth)}: one launch {out[1][0]:.1f} us | three {out[2][0]:.1f} | general {out[0][0]:.1f}   same winner: {out[1][1:] == out[2][1:] == out[0][1:]}", flush=True)

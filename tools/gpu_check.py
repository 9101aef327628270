#!/usr/bin/env python3
"""Stage-by-stage GPU-vs-oracle diagnostics (run on the GPU box): prints max errors per stage."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import gpr, tree
from pygpso_amd import HipGPEngine, _lib as L
from tests.helpers import synthetic_problem, synthetic_leaves

def rel(a, b):
    return float(np.max(np.abs(a - b)) / max(1e-300, np.max(np.abs(b))))

def run(dtype, n, d, m, kernel="Matern52", noise=1e-3, ard=False):
    X, y = synthetic_problem(n, d, seed=0)
    Xs = synthetic_leaves(m, d, seed=1)
    ls = 0.25 * np.sqrt(d) * (np.linspace(0.8, 1.3, d) if ard else np.ones(1))
    th = gpr.Theta(kernel, ls, 1.3, noise, float(y.mean()))
    post = gpr.posterior(th, X, y)
    f_ref, g_ref = gpr.nlml_and_grad(th, X, y)
    eng = HipGPEngine(dtype)
    eng.set_data(X, y)
    t = time.time()
    f, g = eng.fit_eval(kernel, ls, th.variance, th.noise, th.mean_c, want_grad=True)
    t_fit = time.time() - t
    Lg = eng.get_matrix(L.MAT_CHOL); Li = eng.get_matrix(L.MAT_LINV); Ki = eng.get_matrix(L.MAT_KINV)
    al = eng.get_vector(L.VEC_ALPHA)
    Li_ref = np.linalg.inv(post.L)
    print(f"[{dtype} {kernel} n={n} d={d} m={m} ard={ard}] chol {rel(Lg, post.L):.2e} linv {rel(Li, Li_ref):.2e} "
          f"kinv {rel(Ki, Li_ref.T @ Li_ref):.2e} alpha {rel(al, post.alpha):.2e} nlml {abs(f - f_ref) / abs(f_ref):.2e} "
          f"grad {np.max(np.abs(g - g_ref) / np.maximum(1.0, np.abs(g_ref))):.2e} fit_ms(dev) {eng.last_ms(2):.2f} wall {t_fit*1e3:.1f}")
    mean_ref, var_ref = gpr.predict_y(post, Xs)
    t = time.time()
    mean, var = eng.predict(Xs)
    t_pred = time.time() - t
    print(f"    predict: mean err {np.max(np.abs(mean - mean_ref)):.2e} var err {np.max(np.abs(var - var_ref)):.2e} "
          f"(var scale {th.variance}) tile_ms {eng.last_ms(0):.3f} call_ms {eng.last_ms(1):.3f} wall {t_pred*1e3:.1f}")
    vs = gpr.VARSIGMA_DEFAULT
    idx, mu, vv, ucb = eng.best_ucb(Xs, vs)
    ucb_ref = mean_ref + vs * var_ref
    i_ref = int(np.argmax(ucb_ref))
    print(f"    best_ucb: idx {idx[0]} (oracle {i_ref}) ucb {ucb[0]:.12g} (oracle {ucb_ref[i_ref]:.12g}) "
          f"oracle ucb at gpu idx {ucb_ref[idx[0]]:.12g}")
    # set_posterior interop
    eng2 = HipGPEngine(dtype)
    eng2.set_posterior(X, post.L, post.alpha, kernel, ls, th.variance, th.noise, th.mean_c)
    mean2, var2 = eng2.predict(Xs)
    print(f"    set_posterior path: mean err {np.max(np.abs(mean2 - mean_ref)):.2e} var err {np.max(np.abs(var2 - var_ref)):.2e}")

def grow_check():
    eng = HipGPEngine("float64")
    rng = np.random.default_rng(3)
    ok = True
    for d, depth in ((2, 5), (3, 7), (6, 8), (12, 6)):
        b = [(0.0, 1.0)] * d
        # descend a few random splits to get a non-trivial box
        for _ in range(4):
            b = tree.split_bounds(b)[int(rng.integers(3))]
        ref = tree.grow(b, depth)
        got = eng.grow(np.array(b), depth)
        same = np.array_equal(ref, got)
        ok &= same
        print(f"[grow d={d} depth={depth}] rows {got.shape[0]} bit-identical: {same}"
              + ("" if same else f" maxdiff {np.max(np.abs(ref-got)):.3e}"))
    return ok

if __name__ == "__main__":
    print(L.load().gpso_version().decode())
    grow_check()
    run("float64", 50, 2, 121)
    run("float64", 256, 6, 4096)
    run("float64", 300, 5, 1000, kernel="SquaredExponential", ard=True)
    run("float64", 200, 3, 777, kernel="Matern32")
    run("float64", 200, 3, 777, kernel="Matern12")
    run("float32", 256, 6, 4096)
    run("float32", 2048, 12, 65536)

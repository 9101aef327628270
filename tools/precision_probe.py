#!/usr/bin/env python3
"""Accuracy of every engine dtype / predict-math mode in the noise regime the reference runs in
(sigma_n^2 ~ 1.05e-6 at GPflow's floor, sigma^2 ~ 3-7: examples/1-callbacks.ipynb:294-297) against
the float64 oracle.  Run on the GPU box; prints one JSON record per (problem, mode).

Problems: (a) the evaluated points of the G6 run (N = 52, D = 2, produced here by the oracle loop)
at the final theta of that run, leaves = ternary sub-trees of the tree's leaves; (b) synthetic C2 /
C3 shapes at the G6 / G7 thetas with sigma_n^2 = 1.05e-6 (lengthscale also scaled by sqrt(D/2), the
harder, denser regime)."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import gpr, gpso_loop, tree  # noqa: E402
from tests.helpers import load_goldens, rotated_peaks, synthetic_leaves, synthetic_problem  # noqa: E402

VS = gpr.VARSIGMA_DEFAULT


def g6_problem():
    st = gpso_loop.LoopState([(-3, 5), (-3, 3)], depth=5, budget=50)
    gpso_loop.run(st, rotated_peaks)
    ev = [p for p in st.points if p["label"] == gpso_loop.EVALUATED]
    X = np.array([p["coord"] for p in ev])
    y = np.array([p["mu"] for p in ev])
    leaves = np.vstack([tree.grow(n["bounds"], 3) for n in st.preorder() if not n["children"] and n["depth"] >= 3])
    return X, y, st.theta, leaves


def problems():
    G = load_goldens()
    X, y, th, leaves = g6_problem()
    yield "G6-final N=52 D=2", X, y, th, leaves
    for n, d in ((256, 6), (2048, 12)):
        Xs, ys = synthetic_problem(n, d, seed=0)
        for tag, t in (("G7[0]", G["G7"]["theta_after_each_update"][0]), ("G6-final", G["G6"]["final_theta"])):
            for scale in (1.0, float(np.sqrt(d / 2.0))):
                th = gpr.Theta("Matern52", t["lengthscale"] * scale, t["variance"], 1.05e-6, t["mean_c"])
                yield (f"synthetic N={n} D={d} theta={tag} ls x{scale:.2f}", Xs, ys * np.sqrt(t["variance"]), th,
                       synthetic_leaves(4096, d))


def main():
    from pygpso_amd import HipGPEngine

    # dtype:predict_math:generation
    modes = [("float64", "native", "float64"), ("mixed", "native", "float64"), ("mixed", "bf16x6", "float64"),
             ("mixed", "bf16x3", "float64"), ("mixed", "native", "float32"), ("float32", "native", "float64"),
             ("float32", "bf16x6", "float64"), ("float32", "native", "float32")]
    if len(sys.argv) > 1:
        modes = [tuple(m.split(":")) for m in sys.argv[1:]]
    for name, X, y, th, leaves in problems():
        post = gpr.posterior(th, X, y)
        mean_ref, var_ref = gpr.predict_y(post, leaves)
        ucb_ref = mean_ref + VS * var_ref
        i_ref = int(np.argmax(ucb_ref))
        f_ref, g_ref = gpr.nlml_and_grad(th, X, y)
        for dtype, math, gen in modes:
            rec = {"problem": name, "dtype": dtype, "math": math, "gen": gen, "cond_L": float(np.linalg.cond(post.L)),
                   "var_ref_min": float(var_ref.min())}
            try:
                # the self-test is reported, not enforced: this tool measures the raw errors
                eng = HipGPEngine(dtype, predict_math=math, generation=gen if dtype != "float64" else None,
                                  precision_check=False)
                eng.set_data(X, y)
                f, g = eng.fit_eval(th.kernel, th.lengthscales, th.variance, th.noise, th.mean_c, want_grad=True)
                mean, var = eng.predict(leaves)
                idx, mu, vv, uu = eng.best_ucb(leaves, VS)
                rec.update(
                    nlml_rel=abs(f - f_ref) / abs(f_ref),
                    grad_rel=float(np.max(np.abs(g - g_ref) / np.maximum(1.0, np.abs(g_ref)))),
                    dmean=float(np.max(np.abs(mean - mean_ref))),
                    dvar_over_s2=float(np.max(np.abs(var - var_ref)) / th.variance),
                    var_min=float(var.min()),
                    argmax_same=bool(int(idx[0]) == i_ref),
                    ucb_gap_at_winner=float(ucb_ref[i_ref] - ucb_ref[int(idx[0])]),
                    ducb=float(np.max(np.abs(mean + VS * var - ucb_ref))),
                )
                rec["self_test"] = eng.precision_info()
            except Exception as e:  # noqa: BLE001 - report, keep going
                rec["error"] = f"{type(e).__name__}: {e}"
            print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Census for a SCREENED best-UCB call (optional): score every leaf with ONE fp16 piece per operand (one MFMA
per product instead of three), keep the leaves whose UCB could still be the maximum given an error bound, re-score those with
the f16x3 kernel.  Worth building only if few leaves survive.  CPU emulation (numpy): the one-piece arithmetic is emulated by
rounding the power-of-two scaled L^-1 and the generated k* tile to fp16 and accumulating in float32.

Reports, per posterior: the ACTUAL largest UCB error of the one-piece pass (no bound can be smaller), the survivors under that
oracle-knowledge bound (an optimistic floor), and the survivors under a rigorous per-leaf bound
    |d var| <= 2 |A|.|dA| + |dA|^2,  |dA_i| <= 2^-10 sum_j |L^-1_ij| |k*_j|   (needs a second pass with |L^-1|: priced as such).

    python tools/screen_census.py [c3] [--leaves 16384]
"""
import argparse
import json
import math
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import gpr  # noqa: E402
from tests.helpers import synthetic_leaves, synthetic_problem  # noqa: E402

SHAPES = {"c2": (6, 256), "c3": (12, 2048), "c4s": (20, 4096), "n1024d3": (3, 1024)}
ap = argparse.ArgumentParser()
ap.add_argument("shapes", nargs="*", default=["c3"])
ap.add_argument("--leaves", type=int, default=16384)
ap.add_argument("--noise", type=float, nargs="*", default=[1e-3, 1e-6])
a = ap.parse_args()
VS = gpr.VARSIGMA_DEFAULT


def f16_round_scaled(x):
    s = 2.0 ** (13 - math.ceil(math.log2(np.max(np.abs(x)))))
    return (x * s).astype(np.float16).astype(np.float32) / np.float32(s)


for name in a.shapes:
    d, n = SHAPES[name]
    X, y = synthetic_problem(n, d, seed=0)
    Xs = synthetic_leaves(a.leaves, d, seed=1)
    for noise in a.noise:
        th = gpr.Theta("Matern52", np.array([0.25 * math.sqrt(d)]), 1.0, noise, float(y.mean()))
        post = gpr.posterior(th, X, y)
        Linv = np.linalg.inv(post.L)
        Ks = gpr.gram(th.kernel, X, Xs, th.lengthscales, th.variance)  # [n, m]
        A = Linv @ Ks
        mean = Ks.T @ post.alpha + th.mean_c
        var = th.variance + th.noise - np.sum(A * A, axis=0)
        ucb = mean + VS * var
        # one piece per operand, float32 accumulation (the mean keeps its float path: it never went through the split)
        A1 = f16_round_scaled(Linv) @ f16_round_scaled(Ks)
        var1 = th.variance + th.noise - np.sum(A1.astype(np.float64) ** 2, axis=0)
        ucb1 = mean + VS * var1
        err = np.abs(ucb1 - ucb)
        e_max = float(err.max())
        surv_oracle = int(np.sum(ucb1 >= ucb1.max() - 2 * e_max))
        # rigorous per-leaf bound from a second |L^-1| |k*| pass
        dA = 2.0 ** -10 * (np.abs(Linv) @ np.abs(Ks))
        bound = VS * np.sum(2 * np.abs(A) * dA + dA * dA, axis=0)
        lo = ucb1 - bound
        surv_rig = int(np.sum(ucb1 + bound >= lo.max()))
        top = np.sort(ucb)[::-1]
        print(json.dumps({"shape": name, "N": n, "D": d, "leaves": a.leaves, "noise": noise,
                          "ucb_max": float(top[0]), "gap_to_second": float(top[0] - top[1]), "gap_to_100th": float(top[0] - top[99]),
                          "one_piece_max_abs_ucb_error": e_max, "one_piece_median_abs_ucb_error": float(np.median(err)),
                          "survivors_if_the_true_max_error_were_the_bound": surv_oracle,
                          "survivor_fraction_optimistic": surv_oracle / a.leaves,
                          "rigorous_bound_median": float(np.median(bound)), "rigorous_bound_max": float(bound.max()),
                          "survivors_rigorous": surv_rig, "survivor_fraction_rigorous": surv_rig / a.leaves,
                          "same_argmax_one_piece": bool(int(np.argmax(ucb1)) == int(np.argmax(ucb)))}), flush=True)

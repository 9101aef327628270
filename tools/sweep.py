#!/usr/bin/env python3
"""N x D sweep of SURVEY.md section 8(d) on one GPU (float32, Matern-5/2, synthetic data as in bench.py):
posterior-fit ms, NLML+gradient ms (median of 5 after 2 warm-ups, device time) and the leaf-UCB predict
rate with the achieved fraction of the bound of the math in use (f32 MFMA peak, dense bf16 peak / 6 for split-bf16 x6, dense fp16 peak / 3 for the fp16 split).  Prints one JSON line per (N, D)."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygpso_amd import HipGPEngine
from tests.helpers import synthetic_leaves, synthetic_problem

PEAK = 157.3e12
NS = [256, 1024, 2048, 4096, 8192, 16384]
DS = [6, 12, 20, 40]
if len(sys.argv) > 1:
    NS = [int(v) for v in sys.argv[1].split(",")]
if len(sys.argv) > 2:
    DS = [int(v) for v in sys.argv[2].split(",")]


def med(f, reps=5, warm=2):
    for _ in range(warm):
        f()
    return float(np.median([f() for _ in range(reps)]))


for n in NS:
    for d in DS:
        X, y = synthetic_problem(n, d, seed=0)
        ls = np.array([0.25 * np.sqrt(d)])
        c = float(y.mean())
        eng = HipGPEngine("float32")
        eng.set_data(X, y)

        def fit(grad):
            eng.fit_eval("Matern52", ls, 1.0, 1e-3 if n <= 4096 else 1e-2, c, want_grad=grad)
            return eng.last_ms(2)

        post = med(lambda: fit(False))
        grad = med(lambda: fit(True))
        m = 65536 if n <= 4096 else (32768 if n <= 8192 else 16384)
        Xs = synthetic_leaves(m, d, seed=1).astype(np.float32)

        def pred():
            eng.best_ucb(Xs, 1.8213863677184496)
            return eng.last_ms(0)

        kms = med(pred, reps=3, warm=1)
        flops = (float(n) * n + 2.0 * n * d + 20.0 * n) * m
        math = eng.precision_info()["predict_math"]  # what GPSO_MATH_AUTO settled on for this posterior
        bound = {"native": PEAK, "bf16x6": 2500e12 / 6, "bf16x3": 2500e12 / 3, "f16x3": 2500e12 / 3}[math]
        print(json.dumps({"N": n, "D": d, "fit_posterior_ms": round(post, 4), "fit_nlml_grad_ms": round(grad, 4),
                          "leaves": m, "predict_math": math, "leaf_tiles_ms": round(kms, 4),
                          "predictions_per_s_kernel": round(m / (kms * 1e-3)),
                          "algorithmic_tflops": round(flops / (kms * 1e-3) / 1e12, 1),
                          "frac_of_bound": round(flops / (kms * 1e-3) / bound, 4)}), flush=True)
        eng.close()

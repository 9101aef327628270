#!/usr/bin/env python3
"""Repeat the mixed-engine predict of the C2-G6 posterior (tests/test_gpu_precision.py) and report the runs whose
mean leaves its tolerance, with the pattern of the leaves hit (a race shows as whole waves / workgroups)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tests.test_gpu_precision as T
from pygpso_amd import HipGPEngine

name = sys.argv[1] if len(sys.argv) > 1 else "C2-G6"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
X, y, th, leaves, post, mean_ref, var_ref = T._problem(name)
tol = 5e-6 * float(np.max(np.abs(y - th.mean_c))) * max(1.0, float(np.max(np.abs(post.alpha))))
for math in (sys.argv[3].split(",") if len(sys.argv) > 3 else ("bf16x3", "bf16x6", "native")):
    nbad = 0
    for r in range(reps):
        eng = HipGPEngine("mixed", predict_math=math)
        eng.set_data(X, y)
        eng.fit_eval(th.kernel, th.lengthscales, th.variance, th.noise, th.mean_c, want_grad=False)
        mean, var = eng.predict(leaves)
        info = eng.precision_info()
        if r < 4 or len(np.nonzero(np.abs(mean - mean_ref) > tol)[0]):
            print(f"  run {r}: generation {info['generation']} math {info['predict_math']} self-test err mean {info['max_abs_err_mean']:.3e} "
                  f"var {info['max_abs_err_var']:.3e}; leaves: mean err {np.max(np.abs(mean - mean_ref)):.3e} var err {np.max(np.abs(var - var_ref)):.3e}")
        bad = np.nonzero(np.abs(mean - mean_ref) > tol)[0]
        if len(bad):
            nbad += 1
            if nbad <= 12:
                print(f"  {math} run {r}: {len(bad)} leaves off, first {bad[:6]}, last {bad[-1]}, blocks of 32: {[int(b) for b in sorted(set(bad // 32))[:12]]}, lanes {[int(b) % 32 for b in bad[:10]]}, "
                      f"max err {np.max(np.abs(mean - mean_ref)):.2e}, var err {np.max(np.abs(var - var_ref)):.2e}")
    print(f"{name} mixed {math}: {nbad} of {reps} runs with a mean outside {tol:.2e}")

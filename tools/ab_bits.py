#!/usr/bin/env python3
"""Do two builds of libgpso_hip.so give the SAME BITS?  Every library named on the command line predicts the same leaves
from the same posteriors in a fresh process (a list of engine configurations x shapes); the SHA-1 of mean and variance
are printed side by side, differing rows marked.

    python tools/ab_bits.py pygpso_amd/libgpso_hip_prev.so pygpso_amd/libgpso_hip.so
"""
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CASES = [  # dtype, math, generation, contraction (None: leave the default)
    ("float32", "f16x3", "float32", "f32"),
    ("float32", "f16x3", "float32", None),
    ("mixed", "f16x3", "float64", None),
    ("float32", "bf16x6", "float32", None),
    ("float32", "bf16x3", "float32", None),
    ("float32", "native", "float32", None),
    ("float64", None, None, None),
]
SHAPES = [(256, 6, 1000, "Matern52"), (2048, 12, 4096, "Matern52"), (768, 20, 700, "Matern32"), (512, 40, 513, "SquaredExponential")]


def worker():
    sys.path.insert(0, ROOT)
    import numpy as np

    from pygpso_amd import HipGPEngine
    from tests.helpers import synthetic_leaves, synthetic_problem

    out = {}
    for n, d, m, kernel in SHAPES:
        X, y = synthetic_problem(n, d, seed=1)
        Xs = synthetic_leaves(m, d, seed=3)
        for dtype, math, gen, contraction in CASES:
            kw = {}
            if math:
                kw["predict_math"] = math
            if gen:
                kw["generation"] = gen
            eng = HipGPEngine(dtype, precision_check=False, **kw)
            if contraction and hasattr(eng, "set_contraction"):
                eng.set_contraction(contraction)
            eng.set_data(X, y)
            eng.fit_eval(kernel, 0.3 * np.sqrt(d) * np.ones(1), 1.7, 1e-3, float(y.mean()), want_grad=False)
            mean, var = eng.predict(Xs)
            idx, mu, vv, ucb = eng.best_ucb(Xs, 2.0)
            h = hashlib.sha1(mean.tobytes() + var.tobytes() + np.asarray(idx).tobytes() + np.asarray(ucb).tobytes()).hexdigest()[:12]
            out[f"N{n} D{d} {kernel} | {dtype} {math} gen={gen} contraction={contraction}"] = h
            eng.close()
    print("AB_BITS " + json.dumps(out), flush=True)


def main():
    if "--worker" in sys.argv:
        return worker()
    libs = sys.argv[1:]
    res = []
    for lib in libs:
        env = dict(os.environ, GPSO_HIP_LIB=os.path.abspath(lib))
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "--worker"], env=env, capture_output=True, text=True)
        line = [l for l in p.stdout.splitlines() if l.startswith("AB_BITS ")]
        if not line:
            print(lib, "FAILED", p.stderr[-1500:])
            return 1
        res.append(json.loads(line[0][8:]))
    ndiff = 0
    for k in res[0]:
        hs = [r.get(k) for r in res]
        same = all(h == hs[0] for h in hs)
        ndiff += not same
        print(("same  " if same else "DIFF  ") + " ".join(hs) + "  " + k)
    print(f"{ndiff} differing rows of {len(res[0])}")
    return 0


if __name__ == "__main__":
    sys.exit(main())

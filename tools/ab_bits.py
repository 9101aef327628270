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
    # gpso_append (round 6: the same arithmetic in three launches instead of six + two copies): fit N - k, append k, hash what
    # the extended posterior is -- NLML, the factor's new rows, alpha, predictions
    from pygpso_amd import _lib as L

    for n, d, k, dtype in [(2048, 12, 7, "float32"), (2048, 12, 1, "mixed"), (1000, 5, 20, "float64"), (4096, 6, 40, "float32"),
                           (4096, 6, 64, "float64"), (300, 3, 3, "float64"), (8192, 20, 7, "float32")]:
        X, y = synthetic_problem(n, d, seed=2)
        Xs = synthetic_leaves(1500, d, seed=3)
        eng = HipGPEngine(dtype, precision_check=False)
        eng.set_data(X[:n - k], y[:n - k])
        eng.fit_eval("Matern52", 0.25 * np.sqrt(d) * np.ones(1), 1.0, 1e-3, float(y.mean()), want_grad=False)
        eng.predict(Xs[:256])
        f, in_place = eng.append(X[n - k:], y[n - k:])
        mean, var = eng.predict(Xs)
        h = hashlib.sha1(np.float64(f).tobytes() + mean.tobytes() + var.tobytes() + eng.get_vector(L.VEC_ALPHA).tobytes()
                         + (eng.get_matrix(L.MAT_LINV)[n - k:].tobytes() if n <= 4096 else b"")).hexdigest()[:12]
        out[f"append N{n - k}+{k} D{d} {dtype} in_place={in_place}"] = h
        eng.close()
    print("AB_BITS " + json.dumps(out), flush=True)


def main():
    if "--worker" in sys.argv:
        return worker()
    libs = sys.argv[1:]
    res = []
    for lib in libs:
        env = dict(os.environ, GPSO_HIP_LIB=os.path.abspath(lib), GPSO_HIP_LIB_OLDER="1")
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

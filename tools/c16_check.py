#!/usr/bin/env python3
"""The fp16-pipe contraction of the fp16-split predict kernel (GPSO_OPT_CONTRACTION) beside the f32 one, same process,
same posterior: kernel ms of both (alternating), and the error of both against a float64 engine on the same leaves.

    python tools/c16_check.py [N D M ...]      (triples; default: the C3 / C4 / C5-share shapes and two sweep points)
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from pygpso_amd import HipGPEngine  # noqa: E402
from tests.helpers import synthetic_leaves, synthetic_problem  # noqa: E402


def run(n, d, m, steps=30):
    X, y = synthetic_problem(n, d, seed=0)
    theta = ("Matern52", 0.25 * np.sqrt(d) * np.ones(1), 1.0, 1e-3, float(y.mean()))
    leaves64 = synthetic_leaves(m, d)
    leaves = torch.from_numpy(leaves64.astype(np.float32)).cuda()
    ref = HipGPEngine("float64")
    ref.set_data(X, y)
    ref.fit_eval(*theta, want_grad=False)
    msub = min(m, 8192)
    mean_r, var_r = ref.predict(leaves64[:msub].astype(np.float32).astype(np.float64))
    ref.close()
    eng = HipGPEngine("float32", predict_math="f16x3", generation="float32")
    eng.set_precision_check(False)
    eng.set_data(X, y)
    eng.fit_eval(*theta, want_grad=False)
    rec = {"N": n, "D": d, "M": m}
    for _ in range(5):
        eng.best_ucb(leaves, 2.0)
    ks = {"f16": [], "f32": []}
    for it in range(steps):
        for which in ("f16", "f32"):
            eng.set_contraction(which)
            eng.best_ucb(leaves, 2.0)
            eng.best_ucb(leaves, 2.0)
            ks[which].append(eng.last_ms(0))
    for which in ("f16", "f32"):
        eng.set_contraction(which)
        mean, var = eng.predict(leaves64[:msub].astype(np.float32).astype(np.float64))
        rec[which] = {"kernel_ms": float(np.median(ks[which])), "var_err": float(np.max(np.abs(var - var_r))),
                      "mean_err": float(np.max(np.abs(mean - mean_r)))}
    eng.close()
    rec["speedup"] = rec["f32"]["kernel_ms"] / rec["f16"]["kernel_ms"]
    print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    a = [int(v) for v in sys.argv[1:]]
    shapes = [tuple(a[i:i + 3]) for i in range(0, len(a), 3)] or [(2048, 6, 65536), (2048, 12, 65536), (8192, 20, 32768),
                                                                  (16384, 40, 16384), (4096, 40, 65536), (1024, 33, 65536)]
    for s in shapes:
        run(*s)

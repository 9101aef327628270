#!/usr/bin/env python3
"""Kernel time of the split predict kernel at C3 with whatever library GPSO_HIP_LIB names (ablation builds of
csrc/predict.hip: -DGPSO_ABL_HALFGEN / -DGPSO_ABL_NOGEN; their RESULTS are wrong by construction, only the time counts).
Round 4: the -DGPSO_ABL_* hooks live in tools/attic/predict_hooks.patch -- `git apply` it before building the ablation
libraries.  tools/ab_time.py alternates several libraries in one GPU call (boxes differ by up to 10 %)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch  # noqa: E402
from pygpso_amd import HipGPEngine  # noqa: E402
from tests.helpers import synthetic_leaves, synthetic_problem  # noqa: E402

n, d, m = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (2048, 12, 65536)
X, y = synthetic_problem(n, d, seed=0)
leaves = torch.from_numpy(synthetic_leaves(m, d).astype(np.float32)).cuda()
for math in ("f16x3", "bf16x6"):
    eng = HipGPEngine("float32", predict_math=math, generation="float32", precision_check=False)
    eng.set_data(X, y)
    eng.fit_eval("Matern52", 0.25 * np.sqrt(d) * np.ones(1), 1.0, 1e-3, float(y.mean()), want_grad=False)
    ts = []
    for i in range(40):
        eng.best_ucb(leaves, 2.0)
        ts.append(eng.last_ms(0))
    print(os.environ.get("GPSO_HIP_LIB", "shipped"), math, f"kernel ms: median {np.median(ts[10:]):.4f} min {np.min(ts[10:]):.4f}", flush=True)
    eng.close()

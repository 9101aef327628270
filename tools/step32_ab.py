#!/usr/bin/env python3
"""The fused step of the fp16-split kernel on the 32x32x16 matrix instruction ("fused32") beside the one on 16x16x32 ("fused16"):
same posterior, same leaves, one process, alternating -- kernel ms (library events), and how far the two are apart and from a
float64 engine on a sub-sample of the leaves.

    python tools/step32_ab.py [c3 c4 c5 | N D M] [--steps 60]
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CFG = {"c3": (2048, 12, 65536), "c4": (8192, 20, 32768), "c5": (16384, 40, 131072), "c3x4": (2048, 12, 262144)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("shape", nargs="*", default=["c3"])
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--kernel", default="Matern52")
    args = ap.parse_args()
    import torch

    from pygpso_amd import HipGPEngine
    from tests.helpers import synthetic_leaves, synthetic_problem

    shapes = [CFG[s] for s in args.shape] if not args.shape[0].isdigit() else [tuple(map(int, args.shape))]
    for n, d, m in shapes:
        X, y = synthetic_problem(n, d, seed=0)
        theta = (args.kernel, 0.25 * np.sqrt(d) * np.ones(1), 1.0, 1e-3, float(y.mean()))
        leaves_h = synthetic_leaves(m, d).astype(np.float32)
        leaves = torch.from_numpy(leaves_h).cuda()
        eng = HipGPEngine("float32", predict_math="f16x3")
        eng.set_data(X, y)
        eng.fit_eval(*theta, want_grad=False)
        ref = HipGPEngine("float64")
        ref.set_data(X, y)
        ref.fit_eval(*theta, want_grad=False)
        sub = leaves_h[:: max(1, m // 4096)]
        m64, v64 = ref.predict(sub.astype(np.float64))
        ref.close()
        out = {"shape": [n, d, m], "kernel": args.kernel}
        res = {}
        for which in ("fused16", "fused32"):
            eng.set_split_kernel(which)
            mean, var = eng.predict(sub)
            res[which] = (mean, var, eng.best_ucb(leaves, 2.0))
            out[which] = {"max_dmean_vs_f64": float(np.max(np.abs(mean - m64))), "max_dvar_vs_f64": float(np.max(np.abs(var - v64)))}
        out["max_dmean_16_32"] = float(np.max(np.abs(res["fused16"][0] - res["fused32"][0])))
        out["max_dvar_16_32"] = float(np.max(np.abs(res["fused16"][1] - res["fused32"][1])))
        out["same_winner"] = bool(res["fused16"][2][0][0] == res["fused32"][2][0][0])
        ks = {"fused16": [], "fused32": []}
        for r in range(args.rounds):
            for which in ("fused16", "fused32"):
                eng.set_split_kernel(which)
                for _ in range(10):
                    eng.best_ucb(leaves, 2.0)
                t = []
                for _ in range(args.steps):
                    eng.best_ucb(leaves, 2.0)
                    t.append(eng.last_ms(0))
                ks[which].append(float(np.median(t)))
        for which in ks:
            out[which]["kernel_ms_rounds"] = [round(v, 4) for v in ks[which]]
            out[which]["kernel_ms"] = round(float(np.median(ks[which])), 4)
        out["ratio_32_over_16"] = round(out["fused32"]["kernel_ms"] / out["fused16"]["kernel_ms"], 4)
        print(json.dumps(out), flush=True)
        eng.close()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""GPSO_OPT_REUSE on | off in one process, alternating: the default split predict kernels that generate each k-step's cross-Gram
pieces once per workgroup and reload them, beside the kernels that regenerate them per row block -- same bits (checked: means,
variances, winners), kernel ms by the library's events.

    python tools/reuse_ab.py [c3 c4 c5 c3x4 | N D M] [--steps 40] [--rounds 3]
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CFG = {"c3": (2048, 12, 65536), "c4": (8192, 20, 32768), "c5": (16384, 40, 131072), "c3x4": (2048, 12, 262144), "d11": (2048, 12, 118098)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("shape", nargs="*", default=["c3"])
    ap.add_argument("--steps", type=int, default=40)
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
        out = {"shape": [n, d, m], "kernel": args.kernel}
        res = {}
        for on in (0, 1):
            eng.set_reuse(on)
            mean, var = eng.predict(leaves_h)
            best = eng.best_ucb(leaves, 2.0)
            res[on] = (mean.tobytes(), var.tobytes(), tuple(a.tobytes() for a in best))
            out[f"reuse{on}_ran_generate_once"] = eng.last_count(4)
            out[f"reuse{on}_workgroups_per_tile"] = eng.last_count(3)
        out["same_mean_bits"] = res[0][0] == res[1][0]
        out["same_var_bits"] = res[0][1] == res[1][1]
        out["same_winner_bits"] = res[0][2] == res[1][2]
        if not out["same_var_bits"]:
            a, b = np.frombuffer(res[0][1]), np.frombuffer(res[1][1])
            out["max_dvar"] = float(np.max(np.abs(a - b)))
            out["n_diff_var"] = int(np.sum(a != b))
        if not out["same_mean_bits"]:
            a, b = np.frombuffer(res[0][0]), np.frombuffer(res[1][0])
            out["max_dmean"] = float(np.max(np.abs(a - b)))
            out["n_diff_mean"] = int(np.sum(a != b))
        ks = {0: [], 1: []}
        for r in range(args.rounds):
            for on in (0, 1):
                eng.set_reuse(on)
                for _ in range(6):
                    eng.best_ucb(leaves, 2.0)
                t = []
                for _ in range(args.steps):
                    eng.best_ucb(leaves, 2.0)
                    t.append(eng.last_ms(0))
                ks[on].append(float(np.median(t)))
        for on in ks:
            out[f"reuse{on}_kernel_ms"] = round(float(np.median(ks[on])), 4)
            out[f"reuse{on}_kernel_ms_rounds"] = [round(v, 4) for v in ks[on]]
        out["ratio_on_over_off"] = round(out["reuse1_kernel_ms"] / out["reuse0_kernel_ms"], 4)
        print(json.dumps(out), flush=True)
        eng.close()


if __name__ == "__main__":
    main()

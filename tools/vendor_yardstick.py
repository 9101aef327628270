#!/usr/bin/env python3
"""An outside yardstick for the fit and the predict path: the SAME (X, y, theta) through the vendor libraries a
PyTorch-ROCm port of gpso/gp_surrogate.py:490-503 / :313-328 would call -- torch.linalg.cholesky (rocSOLVER potrf),
torch.cholesky_inverse / torch.linalg.solve_triangular (rocBLAS trsm), torch.cdist-free GEMM-form cross-Gram + matmul.
TEST-ONLY TOOL: pygpso_amd never imports torch, rocBLAS or rocSOLVER (SURVEY.md 0 allows them as cross-checks).
HIP-event medians, same box, same run as this library's own numbers:

    python tools/vendor_yardstick.py [c3 c4 c5] > profiles/r05_vendor_yardstick.json
"""
import json
import math
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SHAPES = {"c3": (12, 2048, 65536), "c4": (20, 8192, 32768), "c5": (40, 16384, 131072)}
SQRT5 = math.sqrt(5.0)


def matern52(r2, variance):
    import torch

    r = torch.sqrt(torch.clamp(r2, min=1e-36))
    return variance * (1.0 + SQRT5 * r + (5.0 / 3.0) * r * r) * torch.exp(-SQRT5 * r)


def gram(xa, xb, ls, variance):
    a, b = xa / ls, xb / ls
    r2 = (a * a).sum(1)[:, None] + (b * b).sum(1)[None, :] - 2.0 * a @ b.T
    return matern52(r2, variance)


def timed(fn, reps=5, warm=2):
    import torch

    for _ in range(warm):
        fn()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts)), out


def main():
    import torch

    from pygpso_amd import HipGPEngine
    from tests.helpers import synthetic_leaves, synthetic_problem

    shapes = sys.argv[1:] or ["c3", "c4", "c5"]
    out = {"device": torch.cuda.get_device_name(0), "torch": torch.__version__,
           "note": "torch = what a PyTorch-ROCm port would run (rocSOLVER potrf, rocBLAS trsm / gemm; fit = Gram + Cholesky + "
                   "alpha by two triangular solves, '+inverse' adds L^-1 by a triangular solve against the identity -- what this "
                   "library's posterior holds); ours = gpso_fit_eval (posterior: Gram, Cholesky, L^-1, alpha, NLML, 16-bit pieces) / "
                   "gpso_best_ucb; HIP-event medians of 5 after 2 warm-ups, one process, one box", "cases": []}
    for name in shapes:
        d, n, m = SHAPES[name]
        X, y = synthetic_problem(n, d, seed=0)
        ls, var, noise, c = 0.25 * math.sqrt(d), 1.0, 1e-3, float(y.mean())
        case = {"shape": name, "N": n, "D": d}
        for dt_name, dt in (("float32", torch.float32), ("float64", torch.float64)):
            Xg = torch.from_numpy(X).to("cuda", dt)
            yg = torch.from_numpy(y - c).to("cuda", dt)[:, None]
            eye = torch.eye(n, device="cuda", dtype=dt)

            def fit():
                K = gram(Xg, Xg, ls, var) + noise * eye
                Lc = torch.linalg.cholesky(K)
                a = torch.linalg.solve_triangular(Lc, yg, upper=False)
                alpha = torch.linalg.solve_triangular(Lc.T, a, upper=True)
                return Lc, alpha

            def fit_inv():
                Lc, alpha = fit()
                return torch.linalg.solve_triangular(Lc, eye, upper=False), alpha

            try:
                t_fit, (Lc, alpha) = timed(fit)
                t_chol, _ = timed(lambda: torch.linalg.cholesky(gram(Xg, Xg, ls, var) + noise * eye))
                t_inv, _ = timed(fit_inv)
                case[f"torch_{dt_name}"] = {"fit_ms": t_fit, "gram_plus_cholesky_ms": t_chol, "fit_plus_inverse_ms": t_inv}
            except Exception as exc:  # noqa: BLE001 (a float32 Cholesky may fail at this conditioning: report it)
                case[f"torch_{dt_name}"] = {"error": repr(exc)[:200]}
                continue
            if name == "c3":
                leaves = torch.from_numpy(synthetic_leaves(m, d, seed=1)).to("cuda", dt)

                def predict():
                    Ks = gram(Xg, leaves, ls, var)
                    A = torch.linalg.solve_triangular(Lc, Ks, upper=False)
                    mean = (Ks.T @ alpha)[:, 0] + c
                    v = var + noise - (A * A).sum(0)
                    ucb = mean + 1.8213863677184496 * v
                    return int(torch.argmax(ucb))

                t_pred, _ = timed(predict, reps=3, warm=1)
                case[f"torch_{dt_name}"]["best_ucb_ms"] = t_pred
                case[f"torch_{dt_name}"]["predictions_per_s"] = m / (t_pred * 1e-3)
            del Xg, yg, eye
            torch.cuda.empty_cache()
        for dt_name in ("float32", "float64"):
            if dt_name == "float64" and n > 8192:
                continue
            eng = HipGPEngine(dt_name)
            eng.set_data(X, y)
            ts = []
            for _ in range(7):
                eng.fit_eval("Matern52", ls, var, noise, c, want_grad=False)
                ts.append(eng.last_ms(2))
            ours = {"posterior_fit_ms": float(np.median(ts[2:]))}
            if name == "c3":
                lv = torch.from_numpy(synthetic_leaves(m, d, seed=1).astype(np.float32 if dt_name == "float32" else np.float64)).cuda()
                ks = []
                for _ in range(5):
                    eng.best_ucb(lv, 1.8213863677184496)
                    ks.append(eng.last_ms(1))
                ours["best_ucb_ms"] = float(np.median(ks[2:]))
                ours["predictions_per_s"] = m / (ours["best_ucb_ms"] * 1e-3)
            case[f"ours_{dt_name}"] = ours
            eng.close()
        out["cases"].append(case)
        print(json.dumps(case), file=sys.stderr, flush=True)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()

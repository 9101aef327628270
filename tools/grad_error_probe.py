#!/usr/bin/env python3
"""
Error of the float32 fit's NLML gradient at N = 4096 .. 4600 (two-level factorisation; large products as split-bf16
GEMMs by default, GPSO_OPT_FIT_BF16_SYRK) against the float64 oracle -- beside the SAME computation done by LAPACK in
float32 on the host (spotrf / strtri / sgemm, numpy), which carries nothing but float rounding.  If the two errors
are of one size the tolerance in the tests is the rounding of a float fit (cond x eps), not a kernel's doing.

One JSON line per case: the metric the tests use, max_i |g_i - g_ref_i| / max(1, |g_ref_i|), for the engine with the
bf16 GEMMs on / off and for the LAPACK float32 restatement; cond(K_y), eps x cond, and per-component detail.
"""
import json
import os
import sys

import numpy as np
import scipy.linalg as sla

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import gpr  # noqa: E402
from pygpso_amd import HipGPEngine, _lib as L  # noqa: E402
from tests.helpers import synthetic_problem  # noqa: E402


def lapack_f32_gradient(th, X, y):
    """gpr.nlml_and_grad with the factor, its inverse, K^-1 and alpha computed in float32 by LAPACK / BLAS (kernel
    matrix and the final contractions in float64: the engine generates K in double as well)."""
    r2 = gpr.scaled_sqdist(X, None, th.lengthscales)
    K = gpr.kernel_from_r2(th.kernel, r2, th.variance)
    n = X.shape[0]
    Ky = (K + th.noise * np.eye(n)).astype(np.float32)
    Lf = sla.cholesky(Ky, lower=True, check_finite=False)
    Linv = sla.solve_triangular(Lf, np.eye(n, dtype=np.float32), lower=True, check_finite=False)
    Kinv = (Linv.T @ Linv).astype(np.float64)
    alpha = (Linv.T @ (Linv @ (y - th.mean_c).astype(np.float32))).astype(np.float64)
    W = 0.5 * (Kinv - np.outer(alpha, alpha))
    g = np.empty(4)
    g[0] = np.sum(W * gpr.dk_dlengthscale_iso(th.kernel, r2, K, th.variance, float(th.lengthscales[0])))
    g[1] = np.sum(W * K) / th.variance
    g[2] = np.trace(W)
    g[3] = -np.sum(alpha)
    f = 0.5 * float((y - th.mean_c) @ alpha) + float(np.sum(np.log(np.diag(Lf).astype(np.float64)))) + 0.5 * n * np.log(2 * np.pi)
    return f, g


def metric(g, g_ref):
    return float(np.max(np.abs(g - g_ref) / np.maximum(1.0, np.abs(g_ref))))


def main():
    for n, d in ((4096, 6), (4600, 8)):
        for noise in (1e-2, 1e-3):
            X, y = synthetic_problem(n, d, seed=0)
            th = gpr.Theta("Matern52", 0.25 * np.sqrt(d) * np.ones(1), 1.0, noise, float(y.mean()))
            f_ref, g_ref = gpr.nlml_and_grad(th, X, y)
            r2 = gpr.scaled_sqdist(X, None, th.lengthscales)
            ev = np.linalg.eigvalsh(gpr.kernel_from_r2(th.kernel, r2, th.variance) + noise * np.eye(n))
            cond = float(ev[-1] / ev[0])
            rec = {"n": n, "d": d, "noise": noise, "cond": cond, "eps32_x_cond": float(np.finfo(np.float32).eps * cond),
                   "g_ref": g_ref.tolist(), "f_ref": float(f_ref)}
            f_l, g_l = lapack_f32_gradient(th, X, y)
            rec["lapack_f32"] = {"metric": metric(g_l, g_ref), "abs": np.abs(g_l - g_ref).tolist(), "f_rel": abs(f_l - f_ref) / abs(f_ref)}
            for flag in (1, 0):
                eng = HipGPEngine("float32")
                eng._check(eng._lib.gpso_set_option(eng._h, L.OPT_FIT_BF16_SYRK, flag))
                eng.set_data(X, y)
                f, g = eng.fit_eval(th.kernel, th.lengthscales, th.variance, th.noise, th.mean_c, want_grad=True)
                rec["engine_bf16_gemm" if flag else "engine_f32_gemm"] = {
                    "metric": metric(g, g_ref), "abs": np.abs(g - g_ref).tolist(), "f_rel": abs(f - f_ref) / abs(f_ref)}
                eng.close()
            eng = HipGPEngine("mixed")
            eng.set_data(X, y)
            f, g = eng.fit_eval(th.kernel, th.lengthscales, th.variance, th.noise, th.mean_c, want_grad=True)
            rec["engine_mixed"] = {"metric": metric(g, g_ref), "f_rel": abs(f - f_ref) / abs(f_ref)}
            eng.close()
            print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()

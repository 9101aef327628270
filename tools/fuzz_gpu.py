#!/usr/bin/env python3
"""Randomised GPU-vs-oracle sweep (run on the GPU box): shapes, kernels, dtypes, predict-math
modes, ARD, segmentations.  Exits non-zero on the first violation of the stated tolerances."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import gpr, tree
from pygpso_amd import HipGPEngine
from pygpso_amd._lib import GpsoPrecisionError
from tests.helpers import synthetic_problem, synthetic_leaves

rng = np.random.default_rng(int(os.environ.get("FUZZ_SEED", "1234")))
N_CASES = int(os.environ.get("FUZZ_CASES", "60"))
KERNELS = ["Matern52", "Matern32", "Matern12", "SquaredExponential"]
REPEAT_CASE = int(os.environ.get("FUZZ_REPEAT_CASE", "-1"))
REPEAT = int(os.environ.get("FUZZ_REPEAT", "100"))
bad = 0
t0 = time.time()
refused = 0
for case in range(N_CASES):
    dtype = str(rng.choice(["float64", "float32", "mixed"]))
    n = int(rng.choice([1, 2, 7, 33, 64, 65, 127, 128, 129, 200, 255, 256, 257, 400, 512, 700, 1024]))
    d = int(rng.choice([1, 2, 3, 4, 5, 6, 9, 12, 17, 20, 33, 40, 48]))
    m = int(rng.choice([1, 2, 15, 16, 17, 100, 255, 256, 257, 1000, 4097]))
    kernel = str(rng.choice(KERNELS))
    ard = bool(rng.random() < 0.3) and d > 1
    math = "native"
    if dtype != "float64":
        math = str(rng.choice(["native", "bf16x6", "bf16x3", "f16x3", "auto"]))
    # float engines are exercised down to GPflow's noise floor (1e-6): there they must be right or refuse
    noise = float(rng.choice([1e-6, 1e-5, 1e-4, 1e-3, 1e-2, 1e-1])) if dtype != "float64" else float(rng.choice([1e-6, 1e-4, 1e-2]))
    X, y = synthetic_problem(n, d, seed=int(rng.integers(1 << 30)))
    Xs = synthetic_leaves(m, d, seed=int(rng.integers(1 << 30)))
    ls = 0.25 * np.sqrt(d) * (rng.uniform(0.7, 1.5, size=d) if ard else np.ones(1))
    th = gpr.Theta(kernel, ls, float(rng.uniform(0.5, 2.0)), noise, float(y.mean()) if n > 1 else 0.0)
    try:
        post = gpr.posterior(th, X, y)
    except np.linalg.LinAlgError:
        continue
    f_ref, g_ref = gpr.nlml_and_grad(th, X, y)
    Ky = gpr.gram(kernel, X, None, ls, th.variance) + th.noise * np.eye(n)
    ev = np.linalg.eigvalsh(Ky)
    cond = float(ev[-1] / max(ev[0], 1e-300))
    eng = HipGPEngine(dtype, predict_math=math)
    eng.set_data(X, y)
    tag = (f"{dtype:7s} {math:7s} {kernel:18s} n={n:5d} d={d:2d} m={m:5d} ard={int(ard)} noise={noise:g} cond={cond:.1e}")
    try:
        f, g = eng.fit_eval(kernel, ls, th.variance, th.noise, th.mean_c, want_grad=True)
        mean, var = eng.predict(Xs)
    except (GpsoPrecisionError, np.linalg.LinAlgError) as e:
        # a loud refusal is a correct outcome for the float engines -- but only where the problem is
        # genuinely ill-conditioned for their arithmetic: a float FACTOR from cond ~ 1e3 on (its gate is
        # conservative by design), float applies on a double factor only at extreme conditioning
        limit = 3e2 if dtype == "float32" else 1e7
        # ... and the three-product split (bf16x3, an explicit request: never the default) keeps 2^-16 per product,
        # which the variance's cancellation amplifies: its gate may close from cond ~ 1e4 on (seed 55: 1.4e5,
        # measured variance error 2.1e-4 sigma^2 against the gate's 1.5e-4 -- refused, as it should)
        if math == "bf16x3":
            limit = min(limit, 1e4)
        ok = dtype != "float64" and cond >= limit
        # ... or where the float rounding of the terms k_i alpha_i of the MEAN alone uses up the tolerance
        # (|alpha| grows with the conditioning; the mean is a cancelling sum of such terms)
        ys_ = max(1.0, float(np.max(np.abs(y - th.mean_c))))
        # (a float k* . alpha carries ~eps_float * sum_i |k_i alpha_i| <= eps_float * sigma^2 * sum |alpha_i| of rounding)
        ok = ok or (dtype != "float64" and 6e-8 * float(np.sum(np.abs(post.alpha))) * th.variance >= 0.5e-4 * ys_)
        refused += 1
        bad += (not ok)
        print(f"{'ref' if ok else 'BAD'} {tag} refused: {type(e).__name__} {str(e)[60:330]}")
        continue
    mean_ref, var_ref = gpr.predict_y(post, Xs)
    ys = max(1.0, float(np.max(np.abs(y - th.mean_c))))
    if case == REPEAT_CASE:
        # FUZZ_REPEAT_CASE / FUZZ_REPEAT: run this case's fit + predict again and again (a result that changes
        # between runs is a race, not an accuracy question)
        worst, nbad = 0.0, 0
        for _ in range(REPEAT):
            e2 = HipGPEngine(dtype, predict_math=math)
            e2.set_data(X, y)
            e2.fit_eval(kernel, ls, th.variance, th.noise, th.mean_c, want_grad=False)
            m2, v2 = e2.predict(Xs)
            em = float(np.max(np.abs(m2 - mean_ref))); ev_ = float(np.max(np.abs(v2 - var_ref)))
            worst = max(worst, em)
            nbad += (em > 4e-4 * ys) or (ev_ > 4e-4 * th.variance)
        print(f"REPEAT case {case}: {nbad} of {REPEAT} runs outside the tolerances, worst mean error {worst:.2e}")
    # Tolerances.  float64 and the float64 fit of "mixed": forward errors of a Cholesky-based solve are
    # ~ cond * eps (base 1e-9; Matern-1/2 1e-5).  Float PREDICTIONS that passed the self-test gate: FIXED
    # bounds, 4x the gate's tolerances (1e-4 sigma^2, 1e-4 max|y - c|) -- no conditioning allowance.
    m12 = 30.0 if kernel == "Matern12" else 1.0
    if dtype == "float32":
        e_f = m12 * (5e-5 + cond * 6e-8 * 10.0)
        e_g = m12 * (2e-2 + cond * 6e-8 * 100.0)
    else:
        amp = max(1.0, cond * 2.2e-16 * 1e9)  # forward-error allowance 1.0 * cond * eps once that exceeds 1e-9
        base = 1e-5 if kernel == "Matern12" else 1e-9
        e_f, e_g = base * amp, 1e3 * base * amp
    if dtype == "float64":
        e_m, e_v = base * amp * ys, base * amp * th.variance
    else:
        e_m, e_v = 4e-4 * ys, 4e-4 * th.variance
    errs = dict(nlml=abs(f - f_ref) / max(1.0, abs(f_ref)), grad=float(np.max(np.abs(g - g_ref) / np.maximum(1.0, np.abs(g_ref)))),
                mean=float(np.max(np.abs(mean - mean_ref))), var=float(np.max(np.abs(var - var_ref))))
    ok = errs["nlml"] <= e_f and errs["grad"] <= e_g and errs["mean"] <= e_m and errs["var"] <= e_v
    # segmented best-ucb against numpy on the GPU's own mean/var (exact rule check) ...
    k = int(rng.integers(1, 6))
    cuts = np.sort(rng.integers(0, m + 1, size=k - 1)) if k > 1 else np.array([], dtype=np.int64)
    seg = np.concatenate([[0], cuts, [m]]).astype(np.int64)
    idx, mu, vv, ucb = eng.best_ucb(Xs, gpr.VARSIGMA_DEFAULT, seg)
    full = mean + gpr.VARSIGMA_DEFAULT * var
    for s in range(k):
        a, b = seg[s], seg[s + 1]
        if a == b:
            ok &= idx[s] == -1
        else:
            ok &= (idx[s] == int(np.argmax(full[a:b]))) and (ucb[s] == full[a:b].max())
    # ... and the on-device ternary generator on a random box
    if d <= 12 and case % 3 == 0:
        b = [(0.0, 1.0)] * d
        for _ in range(int(rng.integers(0, 6))):
            b = tree.split_bounds(b)[int(rng.integers(3))]
        depth = int(rng.integers(1, 7))
        ok &= np.array_equal(eng.grow(np.array(b), depth), tree.grow(b, depth))
    # ... and (round 5) gpso_append: k more points at the same hyper-parameters against the oracle's from-scratch posterior
    # of the n + k points, held to the tolerances of the fit above.  Its random draws come from a generator of their own so
    # that the cases of the earlier rounds' seeds stay what they were.
    rng_a = np.random.default_rng(int(os.environ.get("FUZZ_SEED", "1234")) * 1000003 + case)
    if n >= 2 and rng_a.random() < 0.6:
        k_new = int(rng_a.choice([1, 2, 3, 7, 20]))
        X2, y2 = synthetic_problem(k_new, d, seed=int(rng_a.integers(1 << 30)))
        Xa, ya = np.vstack([X, X2]), np.concatenate([y, y2])
        try:
            post_a = gpr.posterior(th, Xa, ya)
            f_a, in_place = eng.append(X2, y2)
            mean_a, var_a = eng.predict(Xs)
            mean_ar, var_ar = gpr.predict_y(post_a, Xs)
            ysa = max(1.0, float(np.max(np.abs(ya - th.mean_c))))
            scale_m = ysa if dtype == "float64" else 1.0
            e_ma = (base * amp * ysa) if dtype == "float64" else 4e-4 * ysa
            errs["app_nlml"] = abs(f_a - post_a.nlml) / max(1.0, abs(post_a.nlml))
            errs["app_mean"] = float(np.max(np.abs(mean_a - mean_ar)))
            errs["app_var"] = float(np.max(np.abs(var_a - var_ar)))
            # (the extended matrix may be worse conditioned than the first: the allowance follows the first's, x 4)
            ok &= errs["app_nlml"] <= 4 * e_f and errs["app_mean"] <= 4 * e_ma and errs["app_var"] <= 4 * e_v
            tag += f" +{k_new}{'' if in_place else 'r'}"
        except (GpsoPrecisionError, np.linalg.LinAlgError):
            tag += f" +{k_new}x"  # (a refusal / loss of positive definiteness after the append: the fit's rule above applies)
            refused += dtype != "float64"
            ok &= dtype != "float64"
    status = "ok " if ok else "BAD"
    bad += (not ok)
    print(f"{status} {tag} " + " ".join(f"{k_}={v:.1e}" for k_, v in errs.items()))
# (round 6) gpso_append where it was built to run: a few cases per sweep at N in [2000, 5000] -- the passes' chunking, the
# 64-wide instantiation, float and double contexts -- against the oracle's from-scratch posterior of the N + k points
N_LARGE = int(os.environ.get("FUZZ_LARGE_APPENDS", "3"))
rng_l = np.random.default_rng(int(os.environ.get("FUZZ_SEED", "1234")) * 7919 + 17)
for case in range(N_LARGE):
    dtype = str(rng_l.choice(["float64", "float32", "mixed"]))
    n = int(rng_l.integers(2000, 5001))
    d = int(rng_l.choice([2, 6, 12, 20, 40]))
    k_new = int(rng_l.choice([1, 3, 7, 20, 33, 64]))
    noise = float(rng_l.choice([1e-3, 1e-2]))
    X, y = synthetic_problem(n + k_new, d, seed=int(rng_l.integers(1 << 30)))
    Xs = synthetic_leaves(700, d, seed=int(rng_l.integers(1 << 30)))
    th = gpr.Theta("Matern52", 0.25 * np.sqrt(d) * np.ones(1), float(rng_l.uniform(0.7, 1.5)), noise, float(y.mean()))
    post_a = gpr.posterior(th, X, y)
    mean_r, var_r = gpr.predict_y(post_a, Xs)
    eng = HipGPEngine(dtype)
    eng.set_data(X[:n], y[:n])
    tag = f"{dtype:7s} large append n={n:5d} +{k_new:2d} d={d:2d} noise={noise:g}"
    try:
        eng.fit_eval(th.kernel, th.lengthscales, th.variance, th.noise, th.mean_c, want_grad=bool(rng_l.random() < 0.5))
        if rng_l.random() < 0.5:
            eng.predict(Xs[:64])  # (pieces built before / after the append)
        f_a, in_place = eng.append(X[n:], y[n:])
        mean_a, var_a = eng.predict(Xs)
    except (GpsoPrecisionError, np.linalg.LinAlgError) as e:
        ok = dtype != "float64"
        refused += 1
        bad += (not ok)
        print(f"{'ref' if ok else 'BAD'} {tag} refused: {type(e).__name__} {str(e)[:200]}")
        continue
    ys = max(1.0, float(np.max(np.abs(y))))
    e_f = 1e-9 if dtype != "float32" else 5e-5
    e_m, e_v = (1e-8 * ys, 1e-8 * th.variance) if dtype == "float64" else (3.3e-4 * ys, 2.4e-5 * th.variance)  # (float: FLOAT_BOUNDS["C4"])
    errs = dict(nlml=abs(f_a - post_a.nlml) / abs(post_a.nlml), mean=float(np.max(np.abs(mean_a - mean_r))), var=float(np.max(np.abs(var_a - var_r))))
    ok = errs["nlml"] <= e_f and errs["mean"] <= e_m and errs["var"] <= e_v
    bad += (not ok)
    print(f"{'ok ' if ok else 'BAD'} {tag} {'in place' if in_place else 'refit'} " + " ".join(f"{k_}={v:.1e}" for k_, v in errs.items()))
print(f"{N_CASES} cases, {refused} refused, {bad} bad, {time.time() - t0:.1f} s")
sys.exit(1 if bad else 0)

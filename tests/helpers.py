"""Shared test helpers: the toy objective, the KAT fixture recipe, seeded synthetic GP problems."""
import json
import math
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_goldens():
    with open(os.path.join(GOLDEN_DIR, "reference_goldens.json")) as fh:
        return json.load(fh)


def rotated_peaks(point):
    """2-D test surface of the GPSO paper (rotated `peaks`), as used by the reference's
    tests/test_optimisation.py:27-42 and notebooks; bounds x in [-3, 5], y in [-3, 3]."""
    px, py = point
    c = np.cos(np.pi / 4)
    s = np.sin(np.pi / 4)
    x = c * px + s * py
    y = c * py - s * px
    return (
        3 * (1 - x) ** 2.0 * np.exp(-(x ** 2) - (y + 1) ** 2)
        - 10 * (x / 5.0 - x ** 3 - y ** 5) * np.exp(-(x ** 2) - y ** 2)
        - 1 / 3 * np.exp(-((x + 1) ** 2) - y ** 2)
    )


def kat_fixture(n_points=10, seed=42):
    """The RNG recipe of tests/test_gp_surrogate.py:221-244 -> list of (coord, mu, sigma, ucb, label)."""
    pts = []
    for i in range(n_points):
        np.random.seed(seed + i)
        coord = np.random.rand(2)
        mu = np.random.rand()
        sigma = np.random.rand()
        ucb = np.random.rand()
        label = int(np.random.choice([1, 2], p=[0.8, 0.2]))
        pts.append((coord, mu, sigma, ucb, label))
    return pts


def kat_training_data():
    pts = kat_fixture()
    x = np.array([p[0] for p in pts if p[4] == 1])
    y = np.array([p[1] for p in pts if p[4] == 1])
    return x, y


def synthetic_problem(n, d, seed=0, noise=1e-2):
    """SURVEY.md section 8(d) synthetic inputs: X ~ U[0,1]^{N x D}, smooth standardised y + noise."""
    rng = np.random.default_rng(seed)
    X = rng.random((n, d))
    w = rng.normal(size=(3, d))
    f = (np.sin(2.0 * X @ w[0]) + np.cos(1.5 * X @ w[1]) * np.exp(-0.5 * np.sum((X - 0.4) ** 2, axis=1))
         + 0.3 * (X @ w[2]))
    if n > 1:
        f = (f - f.mean()) / f.std()
    y = f + noise * rng.normal(size=n)
    return X, y


def synthetic_leaves(m, d, seed=1):
    return np.random.default_rng(seed).random((m, d))


def default_theta_values(d):
    """Throughput-run hyper-parameters of SURVEY.md 8(d): Matern52, l = 0.25 sqrt(D), s2 = 1, noise 1e-3."""
    return dict(kernel="Matern52", lengthscales=0.25 * math.sqrt(d), variance=1.0, noise=1.0e-3)


def g8_run_resume_save_resume(make_optimiser, optimiser_cls, tmp_folder, goldens):
    """The flow of examples/3-saving-resuming-optimisation.ipynb (:193, :406-408, :430) through the drop-in classes:
    run with 25 evaluations, ``resume_run`` with 25 more, ``save_state``, ``resume_from_saved`` with another 25.
    Asserts golden G8: the best point after each leg, evaluation counts and highest scores bit for bit, highest UCB to
    < 1e-8 over the 18 iterations / 77 evaluations."""
    from shutil import rmtree

    g8 = goldens["G8"]
    b = g8["budgets"]
    opt = make_optimiser(g8["depth"], b[0])
    bests = [opt.run(rotated_peaks)]
    bests.append(opt.resume_run(additional_budget=b[1]))
    trace = list(opt.trace)
    opt.save_state(tmp_folder)
    try:
        best3, opt2 = optimiser_cls.resume_from_saved(tmp_folder, additional_budget=b[2], objective_function=rotated_peaks)
    finally:
        rmtree(tmp_folder)
    bests.append(best3)
    trace += list(opt2.trace)
    for best, exp in zip(bests, g8["best_points"]):
        np.testing.assert_almost_equal(best.normed_coord, exp["normed_coord"], decimal=8)
        assert best.score_mu == exp["score_mu"]
    assert [t[0] for t in trace] == [t["evaluations"] for t in g8["trace"]]
    for got, exp in zip(trace, g8["trace"]):
        assert got[1] == exp["highest_score"]
        assert abs(got[2] - exp["highest_ucb"]) < 1e-8
    return opt2


# float32 engines: the winner is the oracle's arg-max, or a leaf whose ORACLE ucb lies within the rounding of a float
# prediction of it (gap x max(1, |ucb_max|)).  Every application of the rule is counted: the GPU run's terminal summary
# (tests/conftest.py) says how often the winner was the oracle's own arg-max and how often the gap branch was needed.
WINNER_CENSUS = {"exact": 0, "gap": 0, "largest_gap_used": 0.0}


def winner_is_the_oracles(idx, ucb_ref, gap, what=""):
    idx = int(idx)
    best = int(np.argmax(ucb_ref))
    if idx == best:
        WINNER_CENSUS["exact"] += 1
        return
    used = float(ucb_ref.max() - ucb_ref[idx]) / max(1.0, abs(float(ucb_ref.max())))
    assert used <= gap, (what, idx, best, used)
    WINNER_CENSUS["gap"] += 1
    WINNER_CENSUS["largest_gap_used"] = max(WINNER_CENSUS["largest_gap_used"], used)

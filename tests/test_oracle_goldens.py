"""
Pins the CPU oracle against every known-answer value the reference holds for the hot path
(G1-G8, tests/golden/reference_goldens.json).  CPU only.
"""
import numpy as np
import pytest

from oracle import gpr, gpso_loop, tree
from tests.helpers import kat_fixture, kat_training_data, load_goldens, rotated_peaks

G = load_goldens()


@pytest.fixture(scope="module")
def kat_post():
    x, y = kat_training_data()
    r = G["fixture_recipe"]
    th0 = gpr.Theta(r["kernel"], r["lengthscale0"], r["variance0"], r["noise0"], r["mean_c0"])
    th = gpr.fit(th0, x, y)
    return gpr.posterior(th, x, y)


def test_G3_fixture_sanity():
    pts = kat_fixture()
    ev = [p for p in pts if p[4] == 1]
    assert len(pts) == G["G3"]["num_points"] and len(ev) == G["G3"]["num_evaluated"]
    best = max(ev, key=lambda p: p[1])
    assert best[1] == G["G3"]["highest_score"]
    np.testing.assert_almost_equal(best[0], G["G3"]["highest_coords"])


def test_G1_predict_y_kat(kat_post):
    mean, var = gpr.predict_y(kat_post, np.array(G["G1"]["predict_at"]))
    assert float(np.around(mean[0], decimals=8)) == G["G1"]["mean"]
    assert float(np.around(var[0], decimals=8)) == G["G1"]["var"]
    # the algebraic shortcut mean = K_*n alpha + c agrees with GPflow's two-solve form
    mean2, var2 = gpr.predict_y_gpflow_order(kat_post, np.array(G["G1"]["predict_at"]))
    np.testing.assert_allclose(mean, mean2, rtol=0, atol=1e-12)
    np.testing.assert_allclose(var, var2, rtol=0, atol=1e-14)


def test_G2_best_ucb_kat(kat_post):
    _, mean, var, ucb = gpr.best_ucb(kat_post, np.array(G["G2"]["predict_at"]))
    exp_ucb = np.around(G["G2"]["mean"] + gpr.VARSIGMA_DEFAULT * G["G2"]["var"], decimals=8)
    assert float(np.around(mean, decimals=8)) == G["G2"]["mean"]
    assert float(np.around(var, decimals=8)) == G["G2"]["var"]
    assert float(np.around(ucb, decimals=8)) == exp_ucb


def test_varsigma_is_erfcinv():
    from scipy.special import erfcinv

    assert gpr.VARSIGMA_DEFAULT == erfcinv(0.01)


def test_analytic_gradient_matches_finite_differences():
    x, y = kat_training_data()
    for kernel in gpr.KERNELS:
        for ls in (np.array([0.3]), np.array([0.3, 0.45])):
            u0 = gpr.Theta(kernel, ls, 1.3, 2e-3, 0.1).pack()
            f0, g0 = gpr.loss_and_grad_unconstrained(kernel, u0, x, y)
            for i in range(u0.shape[0]):
                e = np.zeros_like(u0)
                e[i] = 1e-6
                fp, _ = gpr.loss_and_grad_unconstrained(kernel, u0 + e, x, y)
                fm, _ = gpr.loss_and_grad_unconstrained(kernel, u0 - e, x, y)
                # Matern12 is not differentiable at r = 0: the GEMM-form r^2 on the diagonal is rounding
                # noise (|r^2| ~ 1e-16 -> r ~ 1e-8), which a 1e-6 finite-difference step amplifies
                tol = 5e-2 if kernel == "Matern12" else 2e-5
                assert abs((fp - fm) / 2e-6 - g0[i]) < tol * max(1.0, abs(g0[i])), (kernel, ls, i)


def test_not_positive_definite_raises():
    x = np.array([[0.1, 0.2], [0.1, 0.2], [0.4, 0.4]])
    with pytest.raises(np.linalg.LinAlgError):
        gpr.posterior(gpr.Theta("SquaredExponential", 0.3, 1.0, -1.0e-3, 0.0), x, np.zeros(3))


def _replay(depth, budget):
    st = gpso_loop.LoopState(G["G4"]["bounds"], depth=depth, budget=budget)
    best = gpso_loop.run(st, rotated_peaks)
    return st, best


def test_G4_depth3_budget50():
    st, best = _replay(G["G4"]["depth"], G["G4"]["budget"])
    np.testing.assert_almost_equal(G["G4"]["best_coords"], best["coord"], decimal=G["G4"]["coord_decimals"])
    assert np.around(best["mu"], decimals=8) == G["G4"]["best_score"]


def test_G5_resume_gives_same_answer():
    st = gpso_loop.LoopState(G["G4"]["bounds"], depth=G["G4"]["depth"], budget=G["G5"]["split_budgets"][0])
    gpso_loop.run(st, rotated_peaks)
    best = gpso_loop.resume(st, rotated_peaks, G["G5"]["split_budgets"][1])
    np.testing.assert_almost_equal(G["G4"]["best_coords"], best["coord"], decimal=7)
    assert np.around(best["mu"], decimals=8) == G["G4"]["best_score"]


@pytest.fixture(scope="module")
def g6_state():
    return _replay(G["G6"]["depth"], G["G6"]["budget"])


def test_G6_notebook_trace(g6_state):
    st, best = g6_state
    assert len(st.trace) == len(G["G6"]["trace"]) == 13
    for got, exp in zip(st.trace, G["G6"]["trace"]):
        assert got[0] == exp["evaluations"]
        assert got[1] == exp["highest_score"]  # evaluated scores: bit-for-bit
        assert abs(got[2] - exp["highest_ucb"]) < 1e-9  # through L-BFGS-B: last digits differ by platform
    np.testing.assert_almost_equal(best["coord"], G["G6"]["best"]["normed_coord"], decimal=8)
    assert best["mu"] == G["G6"]["best"]["score_mu"]
    th = st.theta_trace[-1]
    ft = G["G6"]["final_theta"]
    assert f"{th['mean_c']:.6g}" == f"{ft['mean_c']:.6g}"
    assert f"{th['variance']:.6g}" == f"{ft['variance']:.6g}"
    assert f"{th['lengthscales'][0]:.6g}" == f"{ft['lengthscale']:.6g}"
    assert f"{th['noise']:.6g}" == f"{ft['noise']:.6g}"
    # workload facts quoted in BASELINE.md
    assert st.fit_sizes == [5, 6, 7, 9, 12, 15, 19, 23, 27, 31, 36, 40, 45, 52]
    assert st.n_predict_calls == 119 and st.n_leaf_predictions == 13143


def test_G7_theta_after_every_update(g6_state):
    st, _ = g6_state
    exp = G["G7"]["theta_after_each_update"]
    assert len(st.theta_trace) == len(exp) == 14
    for got, e in zip(st.theta_trace, exp):
        for key, val in (("mean_c", got["mean_c"]), ("variance", got["variance"]),
                         ("lengthscale", got["lengthscales"][0]), ("noise", got["noise"])):
            # the notebook prints 6 significant digits
            assert abs(val - e[key]) <= 6e-6 * abs(e[key]), (key, val, e[key])


def test_G8_resume_to_77_evaluations():
    b = G["G8"]["budgets"]
    st = gpso_loop.LoopState(G["G4"]["bounds"], depth=G["G8"]["depth"], budget=b[0])
    best1 = gpso_loop.run(st, rotated_peaks)
    best2 = gpso_loop.resume(st, rotated_peaks, b[1])
    best3 = gpso_loop.resume(st, rotated_peaks, b[2])
    for best, exp in zip((best1, best2, best3), G["G8"]["best_points"]):
        np.testing.assert_almost_equal(best["coord"], exp["normed_coord"], decimal=8)
        assert best["mu"] == exp["score_mu"]
    assert [t[0] for t in st.trace] == [t["evaluations"] for t in G["G8"]["trace"]]
    for got, exp in zip(st.trace, G["G8"]["trace"]):
        assert got[1] == exp["highest_score"]
        assert abs(got[2] - exp["highest_ucb"]) < 1e-8


# ---- ternary geometry (tests/test_param_space.py:130-155,190-206 pin shape and bounds) --------
def test_ternary_split_thirds():
    kids = tree.split_bounds([(0, 1), (0, 1)])
    for i, k in enumerate(kids):
        assert k[0] == (i / 3, (i + 1) / 3) or np.allclose(k[0], (i / 3, (i + 1) / 3))
        assert k[1] == (0, 1)


def test_grow_counts_and_bounds():
    b = [(0.0, 1.0 / 3.0), (0.0, 1.0)]
    c = tree.grow(b, 4)
    assert c.shape == (40, 2) == (tree.grow_count(4), 2)
    assert np.all(c[:, 0] >= b[0][0]) and np.all(c[:, 0] <= b[0][1])
    assert np.all(c[:, 1] >= 0) and np.all(c[:, 1] <= 1)
    # centre child repeats its parent's centre: only 3^(d-1) unique rows
    assert np.unique(c, axis=0).shape[0] == 27


def test_minmax_roundtrip():
    sc = tree.MinMax01([[-3, 5], [-3, 3]])
    x = np.array([[1.0, 0.0], [-3.0, 3.0]])
    np.testing.assert_allclose(sc.inverse_transform(sc.transform(x)), x)
    np.testing.assert_allclose(sc.transform(x), [[0.5, 0.5], [0.0, 1.0]])

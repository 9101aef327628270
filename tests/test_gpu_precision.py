"""
The float-predict engines ("mixed": fit in float64, apply in float32 / split bf16; "float32": all in
float32) in the regime the REFERENCE runs in: GPflow drives the noise variance to its floor on
deterministic objectives -- sigma_n^2 = 1.048e-6 with sigma^2 = 3 .. 7 in every notebook summary
(examples/1-callbacks.ipynb:294-297, examples/0-basic-optimisation.ipynb:327-330) -- i.e.
cond(K + sigma_n^2 I) ~ 1e6 .. 1e7.  Hyper-parameters come from tests/golden/reference_goldens.json
(G6 final theta, G7 theta after the first update).

Stated tolerances (against the float64 oracle)
  mixed, native / bf16x6 / f16x3 (the default ladder's rungs) |d var| <= 5e-6 sigma^2, |d mean| <= 5e-6 max|y - c| * max(1, max|alpha|)
                             identical arg-max, or an oracle UCB within 2e-5 of the oracle's maximum
  mixed, bf16x3 ............ |d var| <= 1e-4 sigma^2 (its 16 mantissa bits), same mean bound
  float32 (float factor) ... with gate tolerances 1e-4: |d var| <= 4e-4 sigma^2, |d mean| <= 4e-4 max|y - c| at
                             the leaves when its self-test passes; otherwise GPSO_E_PRECISION
  every float engine ....... predicted variance > 0 wherever the oracle's is > 2e-5 sigma^2
The G6 run itself (GPSOptimiser, depth 5, budget 50) must reproduce the reference's evaluation counts
6, 7, 9, ..., 52 and its best point in "mixed" arithmetic, and in "float32" either do the same or move
to a more precise engine by itself (never to the CPU).
"""
import logging

import numpy as np
import pytest

from oracle import gpr, gpso_loop, tree
from tests.helpers import load_goldens, rotated_peaks, synthetic_leaves, synthetic_problem

pytestmark = pytest.mark.gpu
G = load_goldens()
VS = gpr.VARSIGMA_DEFAULT
NOISE_FLOOR = 1.05e-6


def _g6_problem():
    """Evaluated points of the G6 run (N = 52, D = 2), its final theta, and the leaves the next
    exploration step would score (ternary sub-trees of the tree's leaves)."""
    st = gpso_loop.LoopState(G["G6"]["bounds"], depth=G["G6"]["depth"], budget=G["G6"]["budget"])
    gpso_loop.run(st, rotated_peaks)
    ev = [p for p in st.points if p["label"] == gpso_loop.EVALUATED]
    X = np.array([p["coord"] for p in ev])
    y = np.array([p["mu"] for p in ev])
    leaves = np.vstack([tree.grow(n["bounds"], 3) for n in st.preorder() if not n["children"] and n["depth"] >= 3])
    return X, y, st.theta, leaves


def _synthetic(n, d, which, scale):
    t = G["G7"]["theta_after_each_update"][0] if which == "G7" else G["G6"]["final_theta"]
    X, y = synthetic_problem(n, d, seed=0)
    th = gpr.Theta("Matern52", t["lengthscale"] * scale, t["variance"], NOISE_FLOOR, t["mean_c"])
    return X, y * np.sqrt(t["variance"]), th, synthetic_leaves(4096, d)


PROBLEMS = {
    "G6": _g6_problem,
    "C2-G7": lambda: _synthetic(256, 6, "G7", 1.0),
    "C2-G7-dense": lambda: _synthetic(256, 6, "G7", np.sqrt(3.0)),
    "C2-G6": lambda: _synthetic(256, 6, "G6", 1.0),
    "C3-G7": lambda: _synthetic(2048, 12, "G7", 1.0),
    "C3-G7-dense": lambda: _synthetic(2048, 12, "G7", np.sqrt(6.0)),
    "C3-G6-dense": lambda: _synthetic(2048, 12, "G6", np.sqrt(6.0)),
}
_cache = {}


def _problem(name):
    if name not in _cache:
        X, y, th, leaves = PROBLEMS[name]()
        post = gpr.posterior(th, X, y)
        mean, var = gpr.predict_y(post, leaves)
        _cache[name] = (X, y, th, leaves, post, mean, var)
    return _cache[name]


def _check(eng, name, tol_var, th, y, leaves, post, mean_ref, var_ref, tol_mean_abs=None):
    mean, var = eng.predict(leaves)
    if tol_mean_abs is None:
        tol_mean_abs = 5e-6 * float(np.max(np.abs(y - th.mean_c))) * max(1.0, float(np.max(np.abs(post.alpha))))
    assert np.max(np.abs(var - var_ref)) <= tol_var * th.variance, (name, np.max(np.abs(var - var_ref)) / th.variance)
    assert np.max(np.abs(mean - mean_ref)) <= tol_mean_abs, (name, np.max(np.abs(mean - mean_ref)), tol_mean_abs)
    assert np.all(var[var_ref > 2e-5 * th.variance] > 0.0)
    ucb_ref = mean_ref + VS * var_ref
    from tests.helpers import winner_is_the_oracles

    winner_is_the_oracles(eng.best_ucb(leaves, VS)[0][0], ucb_ref, 2e-5, name)


@pytest.mark.parametrize("math,tol_var", [("native", 5e-6), ("bf16x6", 5e-6), ("f16x3", 5e-6), ("bf16x3", 1e-4)])
@pytest.mark.parametrize("name", sorted(PROBLEMS))
def test_mixed_engine_at_the_reference_noise_floor(name, math, tol_var):
    from pygpso_amd import HipGPEngine

    X, y, th, leaves, post, mean_ref, var_ref = _problem(name)
    eng = HipGPEngine("mixed", predict_math=math, tol_var=2 * tol_var)
    eng.set_data(X, y)
    f, g = eng.fit_eval(th.kernel, th.lengthscales, th.variance, th.noise, th.mean_c)
    f_ref, g_ref = gpr.nlml_and_grad(th, X, y)
    # the fit is the float64 fit: same tolerances as the float64 parity tests
    assert abs(f - f_ref) <= 1e-9 * abs(f_ref)
    assert np.max(np.abs(g - g_ref) / np.maximum(1.0, np.abs(g_ref))) <= 1e-6
    info = eng.precision_info()
    assert info["passed"], info
    _check(eng, name, tol_var, th, y, leaves, post, mean_ref, var_ref)
    # the self-test (training inputs, closed form) is at least as severe as what the leaves see
    var = eng.predict(leaves)[1]
    assert np.max(np.abs(var - var_ref)) <= max(4.0 * info["max_abs_err_var"], 1e-7 * th.variance)


@pytest.mark.parametrize("name", sorted(PROBLEMS))
def test_float32_engine_is_correct_or_refuses(name):
    from pygpso_amd import HipGPEngine, _lib as L

    X, y, th, leaves, post, mean_ref, var_ref = _problem(name)
    eng = HipGPEngine("float32", tol_var=1e-4, tol_mean=1e-4)
    eng.set_data(X, y)
    try:
        eng.fit_eval(th.kernel, th.lengthscales, th.variance, th.noise, th.mean_c, want_grad=False)
    except np.linalg.LinAlgError:
        return  # a float Cholesky may break down at cond ~ 1e7: a loud refusal as well
    info = eng.precision_info()
    if not info["passed"]:
        with pytest.raises(L.GpsoPrecisionError):
            eng.predict(leaves)
        with pytest.raises(L.GpsoPrecisionError):
            eng.best_ucb(leaves, VS)
        return
    # the gate passed: the leaves are within 4x its tolerances
    _check(eng, name, 4e-4, th, y, leaves, post, mean_ref, var_ref,
           tol_mean_abs=4e-4 * float(np.max(np.abs(y - th.mean_c))))


def test_float32_generation_is_what_breaks_dense_problems():
    """GPflow's GEMM-form r^2 evaluated in float (generation="float32") is the dominant error at the
    reference's lengthscales; the self-test catches it, double generation passes."""
    from pygpso_amd import HipGPEngine, _lib as L

    X, y, th, leaves, post, mean_ref, var_ref = _problem("G6")
    fast = HipGPEngine("mixed", generation="float32")
    fast.set_data(X, y)
    fast.fit_eval(th.kernel, th.lengthscales, th.variance, th.noise, th.mean_c, want_grad=False)
    info = fast.precision_info()
    assert not info["passed"] and info["max_abs_err_var"] > 1e-5 * th.variance
    with pytest.raises(L.GpsoPrecisionError):
        fast.predict(leaves)
    fast.set_precision_check(False)  # the caller may insist
    var = fast.predict(leaves)[1]
    assert np.max(np.abs(var - var_ref)) > 1e-5 * th.variance


def _optimiser(dtype, **engine_options):
    from pygpso_amd import GPRSurrogate, GPSOptimiser, ParameterSpace

    space = ParameterSpace(parameter_names=["x", "y"], parameter_bounds=G["G6"]["bounds"])
    surr = GPRSurrogate.default(dtype=dtype, engine_options=engine_options)
    return GPSOptimiser(parameter_space=space, gp_surrogate=surr, exploration_method="tree",
                        exploration_depth=G["G6"]["depth"], budget=G["G6"]["budget"],
                        stopping_condition="evaluations", update_cycle=1, n_workers=1)


@pytest.mark.parametrize("dtype,math", [("mixed", "native"), ("mixed", "bf16x6"), ("mixed", "f16x3"), ("float32", "native")])
def test_G6_replay_in_float_arithmetic(dtype, math, caplog):
    opt = _optimiser(dtype, predict_math=math)
    with caplog.at_level(logging.WARNING):
        best = opt.run(rotated_peaks)
    assert [t[0] for t in opt.trace] == [t["evaluations"] for t in G["G6"]["trace"]]
    np.testing.assert_almost_equal(best.normed_coord, G["G6"]["best"]["normed_coord"], decimal=7)
    assert best.score_mu == G["G6"]["best"]["score_mu"]  # objective values are exact: same points evaluated
    for got, exp in zip(opt.trace, G["G6"]["trace"]):
        assert got[1] == exp["highest_score"]
        assert abs(got[2] - exp["highest_ucb"]) < 1e-4
    final = opt.gp_surr.gpflow_model.engine.dtype_name
    if final != dtype:  # only ever towards more precision, on the device, and loudly
        assert (dtype, final) in (("float32", "mixed"), ("float32", "float64"), ("mixed", "float64"))
        assert any("reopening the GP posterior" in r.message for r in caplog.records)
    if dtype == "mixed":  # float64 fit: the hyper-parameter path is the reference's to the printed digits
        t, exp = opt.gp_surr.gpflow_model, G["G6"]["final_theta"]
        assert abs(float(t.kernel.lengthscales) - exp["lengthscale"]) <= 6e-6 * exp["lengthscale"]
        assert abs(t.kernel.variance - exp["variance"]) <= 6e-6 * exp["variance"]

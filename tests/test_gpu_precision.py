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


# ---- round 6: a float32 hyper-parameter search that loses positive definiteness (VERDICT r5 missing 2) -----------------------
def test_notpd_inside_a_float32_search_escalates_to_a_mixed_engine_and_ends_where_the_oracle_does(caplog):
    """The reference's search runs in float64 (gpflow.default_float; gpso/gp_surrogate.py:490-503) and does not lose positive
    definiteness for arithmetic reasons.  A float32 factorisation does: on this recipe (N = 4600, D = 8, start (1.3 l*, 1.5,
    3e-3, 0); tools/hyperopt_notpd_probe.py) L-BFGS-B's ninth evaluation fails at pivot 4165.  ``escalate=False`` keeps that
    raise; by default the model reopens its engine as "mixed" (float64 fit) and the search starts over from theta_0.

    Stated tolerances.  Against a float64 ENGINE's search from the same start: the same bits (one history, one arithmetic).
    Against the ORACLE's ``gpr.fit`` (tests/golden/oracle_hyperopt_n4600_d8.json -- 100 CPU evaluations, 159 s, generated
    on the GPU box by the committed probe): NLML within 1e-8 relative; theta within 5e-3 relative -- the optimum sits on the
    noise floor (1e-6) with s2 ~ 6300, cond(K_y) ~ 1e10: a flat valley in which two float64 implementations of the same
    search stop 1.6e-3 apart in s2 (measured) at NLMLs that agree to 1.7e-9.  And the oracle, evaluated HERE at the device's
    theta, must agree with the device's NLML to 1e-8 (measured 2.4e-9: at this conditioning log det and the quadratic form
    each carry ~cond x eps of forward error in ANY float64 implementation) and be no worse than its own optimum by more than 1e-8."""
    import json
    import os

    from pygpso_amd.kernels import Constant, Matern52, Scipy
    from pygpso_amd.model import HipGPR
    from tests.helpers import GOLDEN_DIR

    with open(os.path.join(GOLDEN_DIR, "oracle_hyperopt_n4600_d8.json")) as fh:
        gold = json.load(fh)
    n, d = gold["recipe"]["n"], gold["recipe"]["d"]
    X, y = synthetic_problem(n, d, seed=0)
    st = gold["recipe"]["start"]

    def model(dtype, **kw):
        return HipGPR(data=(X, y[:, None]), kernel=Matern52(lengthscales=st["ls"], variance=st["variance"]),
                      mean_function=Constant(st["c"]), noise_variance=st["noise"], dtype=dtype, **kw)

    plain = model("float32", escalate=False)
    with pytest.raises(np.linalg.LinAlgError, match="pivot"):
        Scipy().minimize(plain.training_loss, plain.trainable_variables)
    assert plain.engine.dtype_name == "float32" and plain.kernel.variance == st["variance"]  # nothing assigned, nothing reopened
    plain.engine.close()

    m = model("float32")
    with caplog.at_level(logging.WARNING):
        res = Scipy().minimize(m.training_loss, m.trainable_variables)
    assert m.engine.dtype_name == "mixed" and m.fit_escalations == 1
    assert "reopening the GP posterior as a 'mixed' engine" in caplog.text
    ref = model("float64")
    res64 = Scipy().minimize(ref.training_loss, ref.trainable_variables)
    assert res.nfev == res64.nfev and np.array_equal(res.x, res64.x) and res.fun == res64.fun
    got = np.array([float(m.kernel.lengthscales), m.kernel.variance, m.likelihood.variance, m.mean_function.c])
    want = np.array(gold["oracle"]["theta"])
    rel = np.abs(got - want) / np.abs(want)
    print(f"escalated float32 search: {res.nfev} evaluations (+ {m.num_loss_evals - res.nfev} abandoned in float32); theta vs the "
          f"oracle's fit: rel {rel}, nlml rel {abs(res.fun - gold['oracle']['nlml']) / abs(gold['oracle']['nlml']):.1e}")
    assert np.all(rel <= 5e-3), (got, want)
    assert abs(res.fun - gold["oracle"]["nlml"]) <= 1e-8 * abs(gold["oracle"]["nlml"])
    th = gpr.Theta("Matern52", np.array([got[0]]), got[1], got[2], got[3])
    f_here = gpr.posterior(th, X, y).nlml  # the oracle at the device's optimum
    assert abs(f_here - res.fun) <= 1e-8 * abs(f_here)
    assert f_here <= gold["oracle"]["nlml"] + 1e-8 * abs(gold["oracle"]["nlml"])
    # the model goes on predicting from the engine it ended on
    Xs = synthetic_leaves(256, d)
    mean, var = m.predict_y(Xs)
    mean_ref, var_ref = gpr.predict_y(gpr.posterior(th, X, y), Xs)
    assert np.max(np.abs(mean[:, 0] - mean_ref)) <= 2e-3 * max(1.0, float(np.max(np.abs(y))))
    assert np.max(np.abs(var[:, 0] - var_ref)) <= 1e-3 * th.variance

"""
The reference's own known-answer tests, run against the HIP path (float64) through the drop-in
classes.  Bodies follow tests/test_gp_surrogate.py:259-347 and tests/test_optimisation.py:66-152
of the reference; expected values come from tests/golden/reference_goldens.json.
"""
import os
from shutil import rmtree

import numpy as np
import pytest

from tests.helpers import g8_run_resume_save_resume, kat_fixture, load_goldens, rotated_peaks

pytestmark = pytest.mark.gpu
G = load_goldens()
TMP = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_tmp_gpu")


def _kat_surrogate():
    from pygpso_amd import GPPoint, GPRSurrogate, PointLabels
    from pygpso_amd import kernels as K

    pts = [GPPoint(*p[:4], PointLabels(p[4])) for p in kat_fixture()]
    return GPRSurrogate(gp_kernel=K.Matern52(), gp_meanf=K.Constant(), points=pts)


def test_G1_gp_train():
    s = _kat_surrogate()
    x, y = s.current_training_data
    s._gp_train(x=x, y=y[:, np.newaxis])
    mean, var = s.gpflow_model.predict_y(np.array([[0.5, 0.5]]))
    assert float(np.around(mean[0, 0], decimals=8)) == G["G1"]["mean"]
    assert float(np.around(var[0, 0], decimals=8)) == G["G1"]["var"]


def test_G1_gp_train_and_predict():
    s = _kat_surrogate()
    x, y = s.current_training_data
    s._gp_train(x=x, y=y[:, np.newaxis])
    at = np.array([[0.5, 0.5]])
    s.gp_predict(at)
    assert len(s.points) == 11
    p = s.points[-1]
    np.testing.assert_equal(p.normed_coord, at.squeeze())
    assert float(np.around(p.score_mu, decimals=8)) == G["G1"]["mean"]
    assert float(np.around(p.score_sigma, decimals=8)) == G["G1"]["var"]


def test_G2_gp_eval_best_ucb():
    s = _kat_surrogate()
    exp_ucb = np.around(G["G2"]["mean"] + s.gp_varsigma * G["G2"]["var"], decimals=8)
    x, y = s.current_training_data
    s._gp_train(x=x, y=y[:, np.newaxis])
    best = s.gp_eval_best_ucb(np.array(G["G2"]["predict_at"]))
    assert float(np.around(best[0], decimals=8)) == G["G2"]["mean"]
    assert float(np.around(best[1], decimals=8)) == G["G2"]["var"]
    assert float(np.around(best[2], decimals=8)) == exp_ucb
    assert len(s.points) == 10


def test_surrogate_save_load_bit_equal_predictions():
    from pygpso_amd import GPRSurrogate

    s = _kat_surrogate()
    x, y = s.current_training_data
    s._gp_train(x=x, y=y[:, np.newaxis])
    s.save(TMP)
    try:
        loaded = GPRSurrogate.from_saved(TMP)
        m1, v1 = s.gpflow_model.predict_y(x)
        m2, v2 = loaded.gpflow_model.predict_y(x)
        np.testing.assert_equal(m1.numpy(), m2.numpy())
        np.testing.assert_equal(v1.numpy(), v2.numpy())
        assert list(s.points) == list(loaded.points)
    finally:
        rmtree(TMP)


def _optimiser(depth, budget):
    from pygpso_amd import GPSOptimiser, ParameterSpace

    space = ParameterSpace(parameter_names=["x", "y"], parameter_bounds=G["G4"]["bounds"])
    return GPSOptimiser(parameter_space=space, exploration_method="tree", exploration_depth=depth,
                        budget=budget, stopping_condition="evaluations", update_cycle=1, n_workers=1)


def test_G4_optimise_v1():
    opt = _optimiser(G["G4"]["depth"], G["G4"]["budget"])
    best = opt.run(rotated_peaks)
    np.testing.assert_almost_equal(G["G4"]["best_coords"], best.normed_coord)
    assert np.around(best.score_mu, decimals=8) == G["G4"]["best_score"]


def test_G5_optimise_resume_and_saved():
    from pygpso_amd import GPSOptimiser

    opt = _optimiser(G["G4"]["depth"], 25)
    opt.run(rotated_peaks)
    opt.save_state(TMP)
    try:
        best = opt.resume_run(additional_budget=25)
        np.testing.assert_almost_equal(G["G4"]["best_coords"], best.normed_coord)
        assert np.around(best.score_mu, decimals=8) == G["G4"]["best_score"]
        best2, _ = GPSOptimiser.resume_from_saved(TMP, additional_budget=25, objective_function=rotated_peaks)
        np.testing.assert_almost_equal(G["G4"]["best_coords"], best2.normed_coord)
        assert np.around(best2.score_mu, decimals=8) == G["G4"]["best_score"]
    finally:
        rmtree(TMP)


def test_G6_G7_notebook_trace_and_hyperparameters():
    thetas = []

    from pygpso_amd import GPSOCallback
    from pygpso_amd.optimisation import CallbackTypes

    class Record(GPSOCallback):
        callback_type = CallbackTypes.post_update

        def run(self, optimiser):
            m = optimiser.gp_surr.gpflow_model
            thetas.append(dict(mean_c=m.mean_function.c, variance=m.kernel.variance,
                               lengthscale=float(m.kernel.lengthscales), noise=m.likelihood.variance))

    opt = _optimiser(G["G6"]["depth"], G["G6"]["budget"])
    opt.callbacks = [Record()]
    best = opt.run(rotated_peaks)
    assert [t[0] for t in opt.trace] == [t["evaluations"] for t in G["G6"]["trace"]]
    for got, exp in zip(opt.trace, G["G6"]["trace"]):
        assert got[1] == exp["highest_score"]  # objective evaluations: bit-for-bit
        assert abs(got[2] - exp["highest_ucb"]) < 1e-8  # through 14 L-BFGS-B fits on another platform
    np.testing.assert_almost_equal(best.normed_coord, G["G6"]["best"]["normed_coord"], decimal=8)
    assert best.score_mu == G["G6"]["best"]["score_mu"]
    assert len(thetas) == len(G["G7"]["theta_after_each_update"])
    for got, exp in zip(thetas, G["G7"]["theta_after_each_update"]):
        for key in ("mean_c", "variance", "lengthscale", "noise"):
            assert abs(got[key] - exp[key]) <= 6e-6 * abs(exp[key]), (key, got[key], exp[key])


def test_G8_run_resume_save_resume_to_77_evaluations():
    """examples/3-saving-resuming-optimisation.ipynb:193,406-408,430 on the HIP engine (float64): 25 evaluations,
    resume_run with 25 more, save_state, GPSOptimiser.resume_from_saved with another 25 -- 18 iterations, 77 evaluations."""
    from pygpso_amd import GPSOptimiser
    from pygpso_amd.engine import HipGPEngine

    opt2 = g8_run_resume_save_resume(_optimiser, GPSOptimiser, TMP, G)
    assert isinstance(opt2.gp_surr.gpflow_model.engine, HipGPEngine)  # the resumed surrogate runs on the device too


# ---- a8: the "sample" exploration method (gpso/optimisation.py:361-364, param_space.py:157-173) ------
@pytest.mark.parametrize("dtype", ["float64", "mixed"])
def test_sample_method_matches_the_oracle_loop(dtype, monkeypatch):
    """exploration_method="sample": every fresh child is scored on depth * D^2 uniform samples of its
    box (np.random.seed(seed) for the first child of a pass only -- the reference's quirk).  The HIP
    engine behind the drop-in classes against the CPU restatement of the reference loop, same seed:
    the evaluation trace is identical (float64: bit for bit; "mixed": float64 fit, float32 predict)."""
    from oracle import gpso_loop
    from pygpso_amd import GPRSurrogate, GPSOptimiser, ParameterSpace

    # np.random.seed(None) -- what every child after the first of a pass gets -- draws OS entropy; make
    # it a reproducible sequence for the two runs being compared
    real_seed = np.random.seed

    def seeding():
        fallback = iter(range(1000, 100000))
        return lambda s=None: real_seed(next(fallback) if s is None else s)

    bounds, depth, budget, seed = G["G4"]["bounds"], 4, 40, 42
    monkeypatch.setattr(np.random, "seed", seeding())
    space = ParameterSpace(parameter_names=["x", "y"], parameter_bounds=bounds)
    opt = GPSOptimiser(parameter_space=space, gp_surrogate=GPRSurrogate.default(dtype=dtype),
                       exploration_method="sample", exploration_depth=depth, budget=budget)
    best = opt.run(rotated_peaks, seed=seed)
    assert opt.max_depth == depth * 2 ** 2
    monkeypatch.setattr(np.random, "seed", seeding())
    st = gpso_loop.LoopState(bounds, depth=depth, budget=budget, method="sample")
    best_ref = gpso_loop.run(st, rotated_peaks, seed=seed)
    assert [t[0] for t in opt.trace] == [t[0] for t in st.trace]
    assert [t[1] for t in opt.trace] == [t[1] for t in st.trace]  # objective values: same points evaluated
    if dtype == "float64":
        tol = 1e-8
    else:  # the float-predict engine's own stated tolerances (its self-test gate), propagated to ucb
        info = opt.gp_surr.gpflow_model.engine.precision_info()
        tol = info["tol_mean_abs"] + opt.gp_surr.gp_varsigma * info["tol_var_abs"]
    assert max(abs(a[2] - b[2]) for a, b in zip(opt.trace, st.trace)) < tol
    np.testing.assert_array_equal(best.normed_coord, best_ref["coord"])
    assert best.score_mu == best_ref["mu"]


# ---- n3: conditional-surrogate grids (gpso/plotting.py:257-381) --------------------------------------
@pytest.mark.parametrize("dtype,tol", [("float64", 1e-9), ("mixed", 5e-6)])
def test_conditional_surrogate_grids_against_per_pair_oracle_predictions(dtype, tol):
    """D = 4, granularity 50: the six 50 x 50 slices through the best point, scored by the HIP predict
    kernel as ONE batch of 15 000 rows, against per-pair oracle predict_y calls (what the reference's
    plot_conditional_surrogate_distributions evaluates pair by pair)."""
    from oracle import gpr
    from pygpso_amd import GPRSurrogate, GPSOptimiser, ParameterSpace, conditional_surrogate_grids

    d, g = 4, 50
    space = ParameterSpace(parameter_names=list("abcd"), parameter_bounds=[[-1, 1]] * d)
    opt = GPSOptimiser(parameter_space=space, gp_surrogate=GPRSurrogate.default(dtype=dtype),
                       exploration_depth=4, budget=30)
    opt.run(lambda p: float(np.exp(-np.sum((np.asarray(p) - 0.2) ** 2)) + 0.1 * np.cos(3 * np.sum(p))))
    grids = conditional_surrogate_grids(opt.gp_surr, granularity=g)
    assert sorted(grids) == [(i, j) for i in range(d) for j in range(i + 1, d)]
    model = opt.gp_surr.gpflow_model
    x, y = opt.gp_surr.current_training_data
    th = gpr.Theta(model.kernel.name, np.atleast_1d(model.kernel.lengthscales), model.kernel.variance,
                   model.likelihood.variance, float(model.mean_function.c))
    post = gpr.posterior(th, x, y)
    best = opt.gp_surr.highest_score.normed_coord
    ax = np.linspace(0, 1, g)
    xg, yg = np.meshgrid(ax, ax)
    scale = max(1.0, float(np.max(np.abs(y))))
    for (i, j), (mean, var) in grids.items():
        at = np.vstack([best] * g * g)
        at[:, i], at[:, j] = xg.flatten(), yg.flatten()
        m_ref, v_ref = gpr.predict_y(post, at)
        assert np.max(np.abs(mean - m_ref.reshape(g, g))) <= tol * scale, (i, j)
        assert np.max(np.abs(var - v_ref.reshape(g, g))) <= tol * th.variance, (i, j)


def test_loss_evaluation_in_the_optimisers_variables():
    """gpso_fit_eval_u (transforms and chain rule inside the library: one C-ABI call per L-BFGS-B evaluation) against
    the same evaluation with the transforms in Python around gpso_fit_eval: the SAME BITS -- loss, gradient, the
    constrained values -- for isotropic, ARD and fixed-mean models, and therefore the same L-BFGS-B iterates: a whole
    hyper-parameter fit ends at the same point after the same number of evaluations."""
    from pygpso_amd import kernels as K
    from pygpso_amd.model import HipGPR
    from tests.helpers import synthetic_problem

    rng = np.random.default_rng(5)
    cases = [(K.Matern52(), K.Constant(), 40, 2), (K.Matern32(lengthscales=[0.3, 0.5, 0.7]), K.Constant(0.2), 90, 3),
             (K.SquaredExponential(lengthscales=0.4), None, 150, 4)]
    for kern, meanf, n, d in cases:
        X, y = synthetic_problem(n, d, seed=n)
        a = HipGPR((X, y[:, None]), kern, meanf, noise_variance=1e-3)
        b = HipGPR((X, y[:, None]), kern, meanf, noise_variance=1e-3)
        b.fused_transforms = False
        u0 = a._pack()
        for _ in range(6):
            u = u0 + rng.normal(size=u0.shape)
            fa, ga = a._loss_and_grad(u)
            fb, gb = b._loss_and_grad(u)
            assert fa == fb and ga.tobytes() == gb.tobytes()
            assert a._device_theta == b._device_theta
    # a whole fit, both ways (the KAT fixture of G1: a degenerate optimum, 30-odd iterations)
    s0 = _kat_surrogate()
    x, y = s0.current_training_data
    res = []
    for fused in (True, False):
        model = HipGPR((x, y[:, np.newaxis]), K.Matern52(), K.Constant(), noise_variance=1e-3)
        model.fused_transforms = fused
        out = K.Scipy().minimize(model.training_loss, model.trainable_variables)
        mean, var = model.predict_y(np.array([[0.5, 0.5]]))
        res.append((model._pack().tobytes(), model.num_loss_evals, out.nit, float(mean[0, 0]), float(var[0, 0])))
        assert float(np.around(mean[0, 0], decimals=8)) == G["G1"]["mean"]
        assert float(np.around(var[0, 0], decimals=8)) == G["G1"]["var"]
    assert res[0] == res[1]

"""
Host-side logic of the drop-in (point store, partition tree, explore/select/update loop, save /
resume), exercised on CPU with the oracle-backed test double standing in for the HIP engine.
The test bodies follow the reference's own tests (tests/test_gp_surrogate.py,
tests/test_param_space.py, tests/test_optimisation.py).
"""
import os
from shutil import rmtree

import numpy as np
import pytest

from oracle import tree as otree
from pygpso_amd import (GPListOfPoints, GPPoint, GPRSurrogate, GPSOptimiser, GPSurrogate, LeafNode,
                        ParameterSpace, PointLabels)
from pygpso_amd import kernels as K
from tests.helpers import kat_fixture, load_goldens, rotated_peaks
from tests.oracle_engine import OracleEngine

G = load_goldens()
TMP = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_tmp_host")


@pytest.fixture(autouse=True)
def oracle_engine(monkeypatch):
    monkeypatch.setattr(GPRSurrogate, "engine_factory", OracleEngine)
    yield
    if os.path.isdir(TMP):
        rmtree(TMP)


# ---- GPListOfPoints (reference tests/test_gp_surrogate.py:25-110) ---------------------------------
def _pt(c, mu=1.0, label=PointLabels.gp_based):
    return GPPoint(np.array(c, dtype=float), mu, 0.5, 0.7, label)


def test_points_append_dedup_rules():
    pts = GPListOfPoints()
    assert pts.append(_pt([0.1, 0.2], label=PointLabels.evaluated)) == 0
    assert pts.append(_pt([0.3, 0.2])) == 1
    assert len(pts) == 2
    # gp_based duplicate is overwritten in place
    assert pts.append(_pt([0.3, 0.2], mu=5.0)) == 1 and len(pts) == 2 and pts[1].score_mu == 5.0
    # evaluated duplicate is never overwritten
    assert pts.append(_pt([0.1, 0.2], mu=9.0)) == 0 and pts[0].score_mu == 1.0
    # tolerance is 1e-12 in L2
    assert pts.find_by_coords(np.array([0.3, 0.2 + 5e-13])) is pts[1]
    assert pts.find_by_coords(np.array([0.3, 0.2 + 2e-12])) is None
    # an evaluated point replaces a gp_based one (what _tree_select does)
    pts.append(_pt([0.3, 0.2], mu=7.0, label=PointLabels.evaluated))
    assert pts[1].label == PointLabels.evaluated and len(pts) == 2
    with pytest.raises(AssertionError):
        pts.append((0.1, 0.2))


def test_points_constructor_does_not_dedup_and_all_duplicates_are_replaced():
    pts = GPListOfPoints([_pt([0.5, 0.5]), _pt([0.5, 0.5]), _pt([0.1, 0.1])])
    assert len(pts) == 3
    pts.append(_pt([0.5, 0.5], mu=3.0))
    assert len(pts) == 3 and pts[0].score_mu == 3.0 and pts[1].score_mu == 3.0


def test_points_cache_survives_list_mutation():
    pts = GPListOfPoints([_pt([0.5, 0.5]), _pt([0.1, 0.1])])
    assert pts.find_index_by_coords(np.array([0.1, 0.1])) == 1
    pts.pop(0)
    assert pts.find_index_by_coords(np.array([0.1, 0.1])) == 0
    pts.insert(0, _pt([0.9, 0.9]))
    assert pts.find_index_by_coords(np.array([0.1, 0.1])) == 1
    for i in range(200):  # growth of the coordinate matrix
        pts.append(_pt([i / 1000.0, 0.77]))
    assert pts.find_index_by_coords(np.array([0.199, 0.77])) == 201


def test_points_save_load_roundtrip():
    os.makedirs(TMP, exist_ok=True)
    pts = GPListOfPoints([GPPoint(*p[:4], PointLabels(p[4])) for p in kat_fixture()])
    pts.save(os.path.join(TMP, "points"))
    loaded = GPListOfPoints.from_file(os.path.join(TMP, "points.json"))
    assert list(loaded) == list(pts)


# ---- GPSurrogate base (reference tests/test_gp_surrogate.py:139-189) --------------------------------
def _fixture_surrogate(cls=GPRSurrogate):
    pts = [GPPoint(*p[:4], PointLabels(p[4])) for p in kat_fixture()]
    return cls(gp_kernel=K.Matern52(), gp_meanf=K.Constant(), points=pts)


def test_G3_properties():
    s = _fixture_surrogate(GPSurrogate)
    assert s.num_evaluated == G["G3"]["num_evaluated"]
    assert s.num_gp_based == G["G3"]["num_points"] - G["G3"]["num_evaluated"]
    hi = s.highest_score
    assert hi.label == PointLabels.evaluated and hi.score_mu == G["G3"]["highest_score"]
    np.testing.assert_almost_equal(hi.normed_coord, G["G3"]["highest_coords"])
    x, y = s.current_training_data
    assert x.shape == (6, 2) and y.shape == (6,)
    assert s.gp_based_coords.shape == (4, 2)
    with pytest.raises(NotImplementedError):
        GPSurrogate.from_saved("")
    with pytest.raises(NotImplementedError):
        s._gp_train(None, None)
    with pytest.raises(NotImplementedError):
        s.save("")


def test_G1_G2_surrogate_interface_with_test_double():
    s = _fixture_surrogate()
    x, y = s.current_training_data
    s._gp_train(x=x, y=y[:, np.newaxis])
    mean, var = s.gpflow_model.predict_y(np.array(G["G1"]["predict_at"]))
    assert float(np.around(mean[0, 0], 8)) == G["G1"]["mean"]
    assert float(np.around(var.numpy()[0, 0], 8)) == G["G1"]["var"]
    n0 = len(s.points)
    best = s.gp_eval_best_ucb(np.array(G["G2"]["predict_at"]))
    assert len(s.points) == n0  # nothing stored
    assert float(np.around(best[0], 8)) == G["G2"]["mean"]
    assert float(np.around(best[1], 8)) == G["G2"]["var"]
    assert float(np.around(best[2], 8)) == np.around(G["G2"]["mean"] + s.gp_varsigma * G["G2"]["var"], 8)
    s.gp_predict(np.array(G["G1"]["predict_at"]))
    assert len(s.points) == n0 + 1
    p = s.points[-1]
    np.testing.assert_equal(p.normed_coord, np.array(G["G1"]["predict_at"][0]))
    assert float(np.around(p.score_mu, 8)) == G["G1"]["mean"]
    assert float(np.around(p.score_sigma, 8)) == G["G1"]["var"]  # the VARIANCE
    assert p.score_ucb == p.score_mu + s.gp_varsigma * p.score_sigma


def test_surrogate_save_load_roundtrip():
    s = _fixture_surrogate()
    x, y = s.current_training_data
    s._gp_train(x=x, y=y[:, np.newaxis])
    s.save(TMP)
    loaded = GPRSurrogate.from_saved(TMP)
    for (k1, v1), (k2, v2) in zip(s.gpflow_model.parameter_dict().items(),
                                  loaded.gpflow_model.parameter_dict().items()):
        assert k1 == k2
        np.testing.assert_allclose(v1, v2)
    m1, v1 = s.gpflow_model.predict_y(x)
    m2, v2 = loaded.gpflow_model.predict_y(x)
    np.testing.assert_equal(m1.numpy(), m2.numpy())
    np.testing.assert_equal(v1.numpy(), v2.numpy())
    assert s.gp_varsigma == loaded.gp_varsigma and s.gp_lik_sigma == loaded.gp_lik_sigma
    assert list(s.points) == list(loaded.points)


# ---- parameter space (reference tests/test_param_space.py) ----------------------------------------
def _space():
    return ParameterSpace(parameter_names=["x", "y"], parameter_bounds=[[-3, 5], [-3, 3]])


def test_space_root_and_normalisation():
    sp = _space()
    assert sp.ndim == 2 and sp.max_depth == 0 and sp.depth == 0 and sp.name == "full_domain"
    assert sp.get_center_as_list(normed=True) == [0.5, 0.5]
    assert sp.get_center_as_list(normed=False) == [1.0, 0.0]
    assert sp.get_center_as_dict(normed=False) == {"x": 1.0, "y": 0.0}
    x = np.array([[1.0, 0.0], [-3.0, 3.0], [0.123, -1.7]])
    np.testing.assert_allclose(sp.denormalise_coords(sp.normalise_coords(x)), x)
    ref = otree.MinMax01([[-3, 5], [-3, 3]])
    np.testing.assert_array_equal(sp.normalise_coords(x), ref.transform(x))
    np.testing.assert_array_equal(sp.denormalise_coords(x), ref.inverse_transform(x))


def test_ternary_split_bounds_and_tree_bookkeeping():
    sp = _space()
    kids = sp.ternary_split()
    assert [k.name for k in kids] == ["full_domain->l", "full_domain->c", "full_domain->r"]
    for i, k in enumerate(kids):
        np.testing.assert_allclose(k.norm_bounds[0], (i / 3, (i + 1) / 3))
        assert tuple(k.norm_bounds[1]) == (0, 1)
        assert k.depth == 1 and k.parent is sp
    assert sp.max_depth == 1 and sp[1] is kids[1]
    grand = kids[0].ternary_split()  # now the second dimension is the widest
    np.testing.assert_allclose(grand[2].norm_bounds[1], (2 / 3, 1.0))
    assert sp.max_depth == 2
    assert [n.name for n in sp.iter_preorder()][:3] == ["full_domain", "full_domain->l", "full_domain->l->l"]


def test_get_best_score_leaf_first_in_preorder_wins_ties():
    sp = _space()
    kids = sp.ternary_split()
    assert sp.get_best_score_leaf(depth=1) is kids[0]  # all scores 0.0 -> first in pre-order
    kids[2].score = 1.0
    kids[1].score = 1.0
    assert sp.get_best_score_leaf(depth=1) is kids[1]
    kids[1].sampled = True
    assert sp.get_best_score_leaf(depth=1) is kids[2]
    assert sp.get_best_score_leaf(depth=1, only_not_sampled=False) is kids[1]
    assert sp.get_best_score_leaf(depth=5) is None
    g1, g2 = kids[2].ternary_split(), kids[0].ternary_split()
    for n in g1 + g2:
        n.score = 3.0
    assert sp.get_best_score_leaf(depth=2) is g2[0]  # child of `l` precedes children of `r`


def test_grow_is_bit_identical_to_reference_recurrence_and_leaves_tree_alone():
    sp = _space()
    node = sp.ternary_split()[2].ternary_split()[0].ternary_split()[1]
    for depth in (0, 1, 4, 7):
        got = node.grow(depth)
        ref = otree.grow(node.norm_bounds, depth)
        assert got.shape == (sum(3 ** i for i in range(depth)), 2)
        np.testing.assert_array_equal(got, ref)
    assert node.children == []
    c = node.grow(4)
    b = node.bounds_array()
    assert np.all(c >= b[:, 0]) and np.all(c <= b[:, 1])


def test_sample_uniformly_in_bounds_and_seeded():
    sp = _space()
    leaf = sp.ternary_split()[0]
    a = leaf.sample_uniformly(50, seed=3)
    assert a.shape == (50, 2) and np.all(a[:, 0] <= 1 / 3) and np.all(a >= 0)
    np.testing.assert_array_equal(a, leaf.sample_uniformly(50, seed=3))


def test_space_pickle_roundtrip():
    os.makedirs(TMP, exist_ok=True)
    sp = _space()
    kids = sp.ternary_split()
    kids[1].score, kids[1].sampled, kids[1].label = 2.5, True, PointLabels.gp_based
    kids[1].ternary_split()
    sp.save(os.path.join(TMP, "space"))
    lo = ParameterSpace.from_file(os.path.join(TMP, "space.pkl"))
    a, b = list(sp.iter_preorder()), list(lo.iter_preorder())
    assert len(a) == len(b) == 7 and lo.max_depth == 2
    for x, y in zip(a, b):
        assert (x.name, x.score, x.sampled, x.label, x.depth) == (y.name, y.score, y.sampled, y.label, y.depth)
        np.testing.assert_array_equal(np.array(x.norm_bounds, dtype=float), np.array(y.norm_bounds, dtype=float))


# ---- the loop (reference tests/test_optimisation.py) -----------------------------------------------
def _optimiser(depth, budget, **kw):
    return GPSOptimiser(parameter_space=_space(), exploration_method="tree", exploration_depth=depth,
                        budget=budget, stopping_condition="evaluations", update_cycle=1, n_workers=1, **kw)


def test_G4_optimise_v1():
    opt = _optimiser(G["G4"]["depth"], G["G4"]["budget"])
    best = opt.run(rotated_peaks)
    np.testing.assert_almost_equal(G["G4"]["best_coords"], best.normed_coord)
    assert np.around(best.score_mu, decimals=8) == G["G4"]["best_score"]
    assert best.label == PointLabels.evaluated


def test_G5_resume_run_and_resume_from_saved():
    opt = _optimiser(G["G4"]["depth"], 25)
    opt.run(rotated_peaks)
    best = opt.resume_run(additional_budget=25)
    np.testing.assert_almost_equal(G["G4"]["best_coords"], best.normed_coord)
    assert np.around(best.score_mu, decimals=8) == G["G4"]["best_score"]

    opt = _optimiser(G["G4"]["depth"], 25)
    opt.run(rotated_peaks)
    opt.save_state(TMP)
    best2, opt2 = GPSOptimiser.resume_from_saved(TMP, additional_budget=25, objective_function=rotated_peaks)
    np.testing.assert_almost_equal(G["G4"]["best_coords"], best2.normed_coord)
    assert np.around(best2.score_mu, decimals=8) == G["G4"]["best_score"]
    assert opt2.n_eval_counter == 55


def test_G8_run_resume_save_resume_through_the_host_classes():
    """Golden G8 (examples/3-saving-resuming-optimisation.ipynb) through the drop-in host classes over the oracle-backed
    engine: the flow tests/test_gpu_goldens.py runs on the HIP engine."""
    from tests.helpers import g8_run_resume_save_resume

    opt2 = g8_run_resume_save_resume(_optimiser, GPSOptimiser, TMP, G)
    assert opt2.n_eval_counter == 77


def test_G6_trace_host_grow_and_device_grow_paths_agree():
    opt = _optimiser(G["G6"]["depth"], G["G6"]["budget"])
    best = opt.run(rotated_peaks)
    assert [t[0] for t in opt.trace] == [t["evaluations"] for t in G["G6"]["trace"]]
    for got, exp in zip(opt.trace, G["G6"]["trace"]):
        assert got[1] == exp["highest_score"] and abs(got[2] - exp["highest_ucb"]) < 1e-9
    assert best.score_mu == G["G6"]["best"]["score_mu"]
    opt_b = _optimiser(G["G6"]["depth"], G["G6"]["budget"])
    opt_b.device_grow = False  # reference-style: host grow + one call per child
    best_b = opt_b.run(rotated_peaks)
    assert opt_b.trace == opt.trace and best_b == best


def test_sample_method_runs():
    opt = GPSOptimiser(parameter_space=_space(), exploration_method="sample", exploration_depth=5,
                       budget=30)
    best = opt.run(rotated_peaks, seed=42)
    assert opt.max_depth == 20 and best.score_mu >= 2.0


def test_eval_repeats_and_saver_protocol():
    saved = []

    class Saver:
        def save_runs(self, results, scores, params):
            saved.append((len(results), len(scores), sorted(params)))

    def obj(p):
        return np.zeros(3), rotated_peaks(p)

    opt = _optimiser(3, 12, saver=Saver())
    best = opt.run(obj, eval_repeats=4)
    assert saved and all(s == (4, 4, ["x", "y"]) for s in saved)
    assert len(saved) == opt.n_eval_counter and best.score_mu > 0


def test_objective_workers_are_spawned_not_forked():
    # n_workers > 1: a spawn pool (a process that holds a HIP context must never fork); the objective
    # has to be importable by the workers, like any multiprocessing target
    opt = _optimiser(3, 10)
    opt.n_workers = 2
    best = opt.run(rotated_peaks, eval_repeats=2)
    ser = _optimiser(3, 10)
    best_ser = ser.run(rotated_peaks, eval_repeats=2)
    assert best == best_ser and opt.trace == ser.trace


def test_conditional_surrogate_grids_match_per_pair_predictions():
    # what gpso/plotting.py:330-356 computes pair by pair, as one batch
    from pygpso_amd import conditional_surrogate_grids

    space = ParameterSpace(parameter_names=["a", "b", "c"], parameter_bounds=[[-1, 1]] * 3)
    opt = GPSOptimiser(parameter_space=space, exploration_depth=3, budget=14)
    opt.run(lambda p: float(np.exp(-np.sum((np.asarray(p) - 0.2) ** 2))))
    g = 7
    grids = conditional_surrogate_grids(opt.gp_surr, granularity=g)
    assert sorted(grids) == [(0, 1), (0, 2), (1, 2)]
    best = opt.gp_surr.highest_score.normed_coord
    ax = np.linspace(0, 1, g)
    xg, yg = np.meshgrid(ax, ax)
    for (i, j), (mean, var) in grids.items():
        at = np.vstack([best] * g * g)
        at[:, i], at[:, j] = xg.flatten(), yg.flatten()
        m_ref, v_ref = opt.gp_surr.gpflow_model.predict_y(at)
        np.testing.assert_allclose(mean, m_ref.numpy().reshape(g, g), rtol=0, atol=1e-12)
        np.testing.assert_allclose(var, v_ref.numpy().reshape(g, g), rtol=0, atol=1e-12)
    assert conditional_surrogate_grids(_fixture_surrogate(GPSurrogate), through=np.array([0.5])) == {}


def test_bad_arguments():
    with pytest.raises(ValueError):
        GPSOptimiser(parameter_space=_space(), exploration_method="nope")
    with pytest.raises(AssertionError):
        GPSOptimiser(parameter_space=_space(), stopping_condition="nope")
    with pytest.raises(AssertionError):
        GPSOptimiser(parameter_space="space")
    with pytest.raises(AssertionError):
        LeafNode(norm_bounds=[(0, 1), (1, 0)], scaler=_space().scaler, parameter_names=["a", "b"])


def test_point_store_index_equals_the_linear_scan():
    """The bucketed index must answer exactly what the reference's linear scan answers
    (gpso/gp_surrogate.py:68-101): L2 distance < 1e-12, all matches in list order -- including pairs
    that straddle a bucket boundary, in-place replacement and growth past the initial capacity."""
    from pygpso_amd.gp_surrogate import DUPLICATE_TOLERANCE, GPListOfPoints

    rng = np.random.default_rng(3)
    width = GPListOfPoints._BUCKET
    pts = GPListOfPoints()
    ref = []  # plain list mirror

    def scan(c):
        return [i for i, q in enumerate(ref) if np.linalg.norm(np.asarray(q) - np.asarray(c)) < DUPLICATE_TOLERANCE]

    def make(c, label=PointLabels.gp_based):
        return GPPoint(normed_coord=np.array(c, dtype=np.float64), score_mu=0.0, score_sigma=0.0, score_ucb=0.0,
                       label=label)

    for step in range(400):
        if step % 5 == 0 and ref:  # a near-duplicate of an existing point, sometimes across a bucket edge
            base = np.array(ref[rng.integers(len(ref))])
            c = base + rng.uniform(-4e-13, 4e-13, size=3)
        elif step % 7 == 0:  # exactly on / next to a bucket boundary
            c = np.array([rng.integers(0, 1000) * width + rng.choice([0.0, 3e-13, -3e-13]), rng.random(), rng.random()])
        else:
            c = rng.random(3)
        want = scan(c)
        got = pts._matches(c)
        assert list(got) == want
        assert pts.find_index_by_coords(c) == (want[0] if want else None)
        idx = pts.append(make(c))
        if want:
            assert idx == want[0]
            for i in want:
                ref[i] = c.copy()  # not evaluated -> replaced, as in the reference
        else:
            assert idx == len(ref)
            ref.append(c.copy())
        assert len(pts) == len(ref)
    assert all(np.array_equal(p.normed_coord, q) for p, q in zip(pts, ref))


def test_direct_lbfgsb_driver_is_scipy_minimize_bit_for_bit():
    """kernels.Scipy drives scipy's own L-BFGS-B routine without the wrappers of scipy.optimize.minimize:
    same iterates, same evaluation count, bit for bit -- on a test function and on a GP loss."""
    import scipy.optimize

    from pygpso_amd.kernels import Scipy

    def rosen(u):
        return scipy.optimize.rosen(u), scipy.optimize.rosen_der(u)

    x0 = np.array([-1.2, 1.0, 0.7, -0.3])
    a = Scipy._lbfgsb_direct(rosen, x0)
    assert a is not None, "scipy.optimize._lbfgsb.setulb changed its signature: the fallback path is in use"
    b = scipy.optimize.minimize(rosen, x0, jac=True, method="L-BFGS-B")
    assert np.array_equal(a.x, b.x) and a.fun == b.fun and (a.nfev, a.nit, a.status) == (b.nfev, b.nit, b.status)

    import types

    s = _fixture_surrogate()
    x, y = s.current_training_data
    s._gp_train(x=x, y=y[:, np.newaxis])  # through Scipy.minimize -> the direct driver
    theta_direct = s.gpflow_model._pack()
    s2 = _fixture_surrogate()
    s2.optimiser = types.SimpleNamespace(minimize=lambda closure, variables=None: closure.__self__._assign(
        scipy.optimize.minimize(closure.__self__._loss_and_grad, closure.__self__._pack(), jac=True,
                                method="L-BFGS-B").x))
    s2._gp_train(x=x, y=y[:, np.newaxis])
    assert np.array_equal(theta_direct, s2.gpflow_model._pack())


def test_refit_every_appends_between_reoptimisations_on_the_host_side():
    """``GPRSurrogate(refit_every=3)`` (opt-in, not the reference's behaviour): updates 1 and 2 after a fit keep the
    hyper-parameters and call ``engine.append`` with exactly the new rows; update 3 re-optimises.  CPU: the engine is the
    oracle-backed test double, so the host logic (what counts as "the same points plus new ones", the counters, the
    model's data) is what is tested; the device arithmetic of gpso_append: tests/test_gpu_append.py."""
    import numpy as np

    from pygpso_amd import GPRSurrogate
    from pygpso_amd.kernels import Constant, Matern52
    from tests.helpers import synthetic_problem
    from tests.oracle_engine import OracleEngine

    calls = []

    class Eng(OracleEngine):
        dtype_name = "float64"

        def append(self, Xn, yn):
            calls.append(np.array(Xn).shape[0])
            return super().append(Xn, yn)

        def set_timing(self, on):
            pass

    X, y = synthetic_problem(40, 2, seed=4)
    surr = GPRSurrogate(gp_kernel=Matern52(lengthscales=0.25, variance=1.0), gp_meanf=Constant(0.0), refit_every=3)
    surr.engine_factory = Eng
    surr.append(X[:30], y[:30])
    surr.gp_update()
    model = surr.gpflow_model
    model.fused_transforms = False  # (the double has no fit_eval_u)
    evals = model.num_loss_evals
    theta = {k: np.array(v) for k, v in model.parameter_dict().items()}
    for step, (lo, hi) in enumerate([(30, 33), (33, 34), (34, 40)], start=1):
        surr.append(X[lo:hi], y[lo:hi])
        surr.gp_update()
        assert model.data[0].shape[0] == hi
        if step < 3:
            assert calls[-1] == hi - lo and model.num_loss_evals == evals
            assert all(np.array_equal(theta[k], v) for k, v in model.parameter_dict().items())
        else:
            assert len(calls) == 2 and model.num_loss_evals > evals
    # the default surrogate never appends
    plain = GPRSurrogate(gp_kernel=Matern52(lengthscales=0.25, variance=1.0), gp_meanf=Constant(0.0))
    plain.engine_factory = Eng
    plain.append(X[:30], y[:30]); plain.gp_update()
    plain.gpflow_model.fused_transforms = False
    plain.append(X[30:33], y[30:33]); plain.gp_update()
    assert len(calls) == 2


# ---- round 6: a float32 hyper-parameter search that loses positive definiteness escalates (host logic) -------------------------
class _FloatEngineThatLosesPD(OracleEngine):
    """A "float32" engine whose factorisation fails below a noise level -- what the device's float Cholesky does at large N when
    L-BFGS-B's line search steps into small noise; the "mixed" engine it is replaced by never fails."""
    dtype_name = "float32"
    opened = []

    def __init__(self, dtype="float32", device=0, **kw):
        super().__init__()
        self.dtype_name = dtype
        self.closed = False
        type(self).opened.append(self)

    def set_timing(self, on):
        pass

    def close(self):
        self.closed = True

    def fit_eval(self, kernel, lengthscales, variance, noise, mean_c, want_grad=True):
        if self.dtype_name == "float32" and noise < 5.0e-3:
            raise np.linalg.LinAlgError("K + noise*I is not positive definite: Cholesky failed at pivot 17")
        return super().fit_eval(kernel, lengthscales, variance, noise, mean_c, want_grad)


def _model_on(engine_cls, monkeypatch, **kw):
    from pygpso_amd.kernels import Constant, Matern52
    from pygpso_amd.model import HipGPR
    from tests.helpers import synthetic_problem

    engine_cls.opened = []
    monkeypatch.setattr(HipGPR, "_open_engine", lambda self, dtype: engine_cls(dtype))
    X, y = synthetic_problem(60, 2, seed=6)
    model = HipGPR(data=(X, y[:, None]), kernel=Matern52(lengthscales=0.4, variance=1.5), mean_function=Constant(0.0),
                   noise_variance=3.0e-2, dtype="float32", **kw)
    model.fused_transforms = False
    return model, X, y


def test_notpd_inside_a_float32_search_reopens_the_engine_as_mixed_and_restarts(monkeypatch, caplog):
    from oracle import gpr
    from pygpso_amd.kernels import Scipy

    model, X, y = _model_on(_FloatEngineThatLosesPD, monkeypatch)
    theta0 = gpr.Theta("Matern52", np.array([0.4]), 1.5, 3.0e-2, 0.0)
    want, info = gpr.fit(theta0, X, y, return_info=True)  # the reference's float64 search from the same start
    assert want.noise < 5.0e-3  # (the search's path crosses the level at which the float engine fails)
    with caplog.at_level("WARNING"):
        res = Scipy().minimize(model.training_loss, model.trainable_variables)
    first, second = _FloatEngineThatLosesPD.opened
    assert first.dtype_name == "float32" and first.closed and second.dtype_name == "mixed" and model.engine is second
    assert model.fit_escalations == 1 and "reopening the GP posterior as a 'mixed' engine" in caplog.text
    # one history in one arithmetic: the restarted search IS the float64 search from theta_0 -- same iterates, same optimum
    # (the model's chain rule and the oracle's differ in the last bits: the iterates agree to ~1e-8)
    assert res.nfev == info.nfev and np.allclose(res.x, info.x, rtol=1e-6, atol=1e-6)
    assert abs(model.likelihood.variance - want.noise) <= 1e-6 * want.noise and abs(model.kernel.variance - want.variance) <= 1e-6 * want.variance
    want = gpr.Theta("Matern52", np.atleast_1d(model.kernel.lengthscales), model.kernel.variance, model.likelihood.variance,
                     model.mean_function.c)
    # ... and the model predicts from the new engine
    mean, var = model.predict_y(X[:4])
    mean_ref, var_ref = gpr.predict_y(gpr.posterior(want, X, y), X[:4])
    assert np.allclose(mean[:, 0], mean_ref, atol=1e-12) and np.allclose(var[:, 0], var_ref, atol=1e-12)


def test_notpd_without_escalation_or_on_a_float64_engine_is_the_callers(monkeypatch):
    from pygpso_amd.kernels import Scipy

    model, X, y = _model_on(_FloatEngineThatLosesPD, monkeypatch, escalate=False)
    with pytest.raises(np.linalg.LinAlgError, match="pivot 17"):
        Scipy().minimize(model.training_loss, model.trainable_variables)
    assert len(_FloatEngineThatLosesPD.opened) == 1 and model.kernel.variance == 1.5  # nothing reopened, nothing assigned

    class F64ThatFails(_FloatEngineThatLosesPD):
        def fit_eval(self, *a, **kw):
            raise np.linalg.LinAlgError("K + noise*I is not positive definite: Cholesky failed at pivot 3")

    model, X, y = _model_on(F64ThatFails, monkeypatch)
    model.engine.dtype_name = "mixed"  # a float64 FIT that fails has nowhere to go: as in the reference, the error escapes
    with pytest.raises(np.linalg.LinAlgError, match="pivot 3"):
        Scipy().minimize(model.training_loss, model.trainable_variables)
    # an engine the model does not own is never replaced
    eng = _FloatEngineThatLosesPD("float32")
    from pygpso_amd.kernels import Constant, Matern52
    from pygpso_amd.model import HipGPR

    m2 = HipGPR(data=(X, y[:, None]), kernel=Matern52(lengthscales=0.4, variance=1.5), mean_function=Constant(0.0),
                noise_variance=3.0e-2, engine=eng)
    m2.fused_transforms = False
    with pytest.raises(np.linalg.LinAlgError):
        Scipy().minimize(m2.training_loss, m2.trainable_variables)
    assert m2.engine is eng


def test_a_fit_at_the_stored_hyperparameters_escalates_too(monkeypatch):
    """``_ensure_resident`` (a predict after ``model.data = ...`` / after loading a saved surrogate) fits at the stored
    hyper-parameters: a float32 engine that cannot factorise there is replaced the same way."""
    model, X, y = _model_on(_FloatEngineThatLosesPD, monkeypatch)
    model.likelihood.variance = 1.0e-3
    mean, var = model.predict_y(X[:4])
    assert model.engine.dtype_name == "mixed" and np.all(np.isfinite(mean))


def test_a_failed_append_leaves_model_and_surrogate_in_one_state():
    """``append_data`` claimed N + k points and a resident posterior BEFORE the device call; a failing append
    left the model describing something the device did not hold."""
    from pygpso_amd.kernels import Constant, Matern52
    from tests.helpers import synthetic_problem

    class Eng(OracleEngine):
        dtype_name = "float64"
        fail = False

        def append(self, Xn, yn):
            if type(self).fail:
                raise np.linalg.LinAlgError("the appended block's Cholesky failed at pivot 31")
            return super().append(Xn, yn)

        def set_timing(self, on):
            pass

    X, y = synthetic_problem(40, 2, seed=4)
    surr = GPRSurrogate(gp_kernel=Matern52(lengthscales=0.25, variance=1.0), gp_meanf=Constant(0.0), refit_every=5)
    surr.engine_factory = Eng
    surr.append(X[:30], y[:30])
    surr.gp_update()
    model = surr.gpflow_model
    model.fused_transforms = False
    evals = model.num_loss_evals
    Eng.fail = True
    with pytest.raises(np.linalg.LinAlgError):
        model.append_data(X[30:32], y[30:32])
    assert model.data[0].shape[0] == 32 and not model._resident and model._device_theta is None
    assert model.engine.n == 32  # ... and the engine was handed exactly that data
    model.data = (X[:30], y[:30, None])
    # through the surrogate: the update re-optimises instead (counted as an update, the points are all there)
    updates = surr._updates
    surr.append(X[30:33], y[30:33])
    surr.gp_update()
    assert model.num_loss_evals > evals and model.data[0].shape[0] == 33 and surr._updates == updates + 1
    Eng.fail = False
    evals = model.num_loss_evals
    surr.append(X[33:35], y[33:35])
    surr.gp_update()  # the next update appends again
    assert model.num_loss_evals == evals and model.data[0].shape[0] == 35


def test_refit_guard_turns_a_surprising_append_into_a_reoptimisation():
    """nothing said how far theta may drift before an appended posterior is stale.  The append's NLML
    increment per new point is -log p(y_new | data, theta): new points far off what theta predicts (here: the function's
    scale changes) make THIS update re-optimise; ordinary points do not."""
    from pygpso_amd.kernels import Constant, Matern52
    from tests.helpers import synthetic_problem

    class Eng(OracleEngine):
        dtype_name = "float64"

        def set_timing(self, on):
            pass

    X, y = synthetic_problem(60, 2, seed=4)
    surr = GPRSurrogate(gp_kernel=Matern52(lengthscales=0.25, variance=1.0), gp_meanf=Constant(0.0), refit_every=10,
                        refit_guard=2.0)
    surr.engine_factory = Eng
    surr.append(X[:40], y[:40])
    surr.gp_update()
    model = surr.gpflow_model
    model.fused_transforms = False
    evals = model.num_loss_evals
    surr.append(X[40:44], y[40:44])  # points of the same function: appended at the kept hyper-parameters
    surr.gp_update()
    assert model.num_loss_evals == evals and surr.guard_refits == 0
    surr.append(X[44:48], y[44:48] * 30.0 + 5.0)  # the objective's scale jumps: 4 points the posterior calls impossible
    surr.gp_update()
    assert surr.guard_refits == 1 and model.num_loss_evals > evals and model.data[0].shape[0] == 48
    # without a guard the same update appends
    plain = GPRSurrogate(gp_kernel=Matern52(lengthscales=0.25, variance=1.0), gp_meanf=Constant(0.0), refit_every=10,
                         refit_guard=None)
    plain.engine_factory = Eng
    plain.append(X[:44], y[:44])
    plain.gp_update()
    plain.gpflow_model.fused_transforms = False
    e0 = plain.gpflow_model.num_loss_evals
    plain.append(X[44:48], y[44:48] * 30.0 + 5.0)
    plain.gp_update()
    assert plain.gpflow_model.num_loss_evals == e0 and plain.guard_refits == 0


def test_refit_every_appends_although_an_evaluation_overwrote_a_gp_based_point_in_place():
    """An evaluated point that duplicates a stored gp-based point overwrites that entry IN PLACE (the reference's rule,
    gpso/gp_surrogate.py:68-101), so the evaluated points of the list do not only grow at its end.  Round 5's
    ``refit_every`` compared prefixes and silently re-optimised on most updates of a real run (9 appends of 31 updates at
    refit_every = 5); the model's points are now matched as a SET and the others appended."""
    from pygpso_amd.kernels import Constant, Matern52
    from tests.helpers import synthetic_problem

    class Eng(OracleEngine):
        dtype_name = "float64"

        def set_timing(self, on):
            pass

    X, y = synthetic_problem(40, 2, seed=4)
    surr = GPRSurrogate(gp_kernel=Matern52(lengthscales=0.25, variance=1.0), gp_meanf=Constant(0.0), refit_every=4)
    surr.engine_factory = Eng
    surr.append(X[:30], y[:30])
    surr.gp_update()
    model = surr.gpflow_model
    model.fused_transforms = False
    evals = model.num_loss_evals
    surr.gp_predict(X[30:33])  # three gp-based points enter the list BEHIND the 30 evaluated ones ...
    surr.append(X[33:35], y[33:35])  # ... two evaluated points behind those ...
    surr.append(X[30:31], y[30:31])  # ... and the first gp-based point is evaluated: overwritten in place, in FRONT of them
    xs, _ = surr.current_training_data
    assert np.array_equal(xs[30], X[30]) and np.array_equal(xs[31], X[33])  # not a prefix extension of the model's data
    surr.gp_update()
    assert model.num_loss_evals == evals  # appended at the kept hyper-parameters all the same
    assert model.data[0].shape[0] == 33 and np.array_equal(model.data[0][:30], X[:30])
    assert sorted(map(tuple, model.data[0][30:].tolist())) == sorted(map(tuple, X[[30, 33, 34]].tolist()))
    # a changed score of an old point is not an append
    surr.points[0] = surr.points[0]._replace(score_mu=surr.points[0].score_mu + 1.0)
    surr.append(X[35:36], y[35:36])
    surr.gp_update()
    assert model.num_loss_evals > evals


def test_point_store_lookup_does_not_degenerate_on_tree_points_and_keeps_the_tolerance_rule():
    """The centres of a ternary tree share coordinates: hashed on coordinate 0 alone (rounds 1-5) nearly every point of a
    D = 12 run fell into one bucket and a look-up scanned the list (250 us at 1 000 points, O(P^2) per gp_update).  The
    store now hashes a projection of all coordinates; the duplicate rule (L2 distance < 1e-12) is unchanged."""
    import time

    rng = np.random.default_rng(0)
    d, n = 12, 3000
    grid = rng.integers(0, 27, size=(n, d)) / 27.0 + 1.0 / 54.0
    grid[:, 0] = 0.5  # every point shares coordinate 0 (and most share several others)
    grid = np.unique(grid, axis=0)
    pts = GPListOfPoints()
    for row in grid:
        pts.append(_pt(row))
    assert len(pts) == grid.shape[0]
    t0 = time.perf_counter()
    for i in range(0, grid.shape[0], 3):
        assert pts.find_index_by_coords(grid[i]) == i
    per = (time.perf_counter() - t0) / (grid.shape[0] / 3)
    assert per < 100e-6, f"{per * 1e6:.0f} us per look-up"  # (a linear scan of 3 000 rows of 12 numbers in Python: > 2 ms)
    assert max(len(v) for v in pts._buckets.values()) <= 3
    # the tolerance rule at random places, bucket edges of the projection included
    for _ in range(300):
        i = int(rng.integers(grid.shape[0]))
        delta = rng.normal(size=d)
        delta /= np.linalg.norm(delta)
        assert pts.find_index_by_coords(grid[i] + 0.9e-12 * delta) == i
        assert pts.find_index_by_coords(grid[i] + 3.0e-12 * delta) is None
    w = np.array(GPListOfPoints._W[:d])
    base = np.full(d, 0.25)
    base[0] += (np.ceil(w @ base / 1e-6) * 1e-6 - w @ base) / w[0] - 2e-13 / w[0]  # projection 2e-13 below a bucket edge
    edge = GPListOfPoints([_pt(base)])
    assert edge.find_index_by_coords(base + 0.6e-12 * w / np.linalg.norm(w)) == 0  # (the duplicate sits in the NEXT bucket)


def test_gp_update_by_index_equals_the_references_reappending():
    """gp_update re-predicts every gp-based point; the reference re-appends them (each overwrites its own entry after a
    look-up), the drop-in updates them by index while the store can vouch for uniqueness.  Same run either way: trace,
    points, scores."""
    from pygpso_amd.gp_surrogate import GPListOfPoints as Store

    def run(force_reference_way):
        space = ParameterSpace(parameter_names=["x", "y"], parameter_bounds=[[-3, 5], [-3, 3]])
        opt = GPSOptimiser(space, exploration_depth=4, budget=40)
        if force_reference_way:
            real = Store.update_scores
            Store.update_scores = lambda self, *a, **k: False
        try:
            best = opt.run(rotated_peaks)
        finally:
            if force_reference_way:
                Store.update_scores = real
        pts = [(tuple(p.normed_coord), p.score_mu, p.score_sigma, p.score_ucb, p.label) for p in opt.gp_surr.points]
        return opt.trace, pts, best.score_mu, opt.gp_surr.points._unique

    fast, slow = run(False), run(True)
    assert fast[3] and fast[0] == slow[0] and fast[2] == slow[2]
    assert fast[1] == slow[1]

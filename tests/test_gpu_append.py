"""
GPU parity tests of ``gpso_append`` (the rank-k append at FIXED hyper-parameters): the posterior
extended in place on the device against the ORACLE's from-scratch posterior on the N + k points.

Stated tolerances
  float64 ..... mean, var, NLML, L, L^-1, alpha <= 1e-9 relative (the tolerances of the from-scratch fit)
  mixed ....... factor as float64; predictions in the float class of tests/test_gpu_parity.py (SMALL_FLOAT_BOUNDS)
  float32 ..... the float-fit bounds of the from-scratch fit: NLML 2e-5 relative, predictions FLOAT_BOUNDS-class
"""
import numpy as np
import pytest

from oracle import gpr
from tests.helpers import synthetic_leaves, synthetic_problem

pytestmark = pytest.mark.gpu

VS = gpr.VARSIGMA_DEFAULT


def _engine(dtype="float64", **kw):
    from pygpso_amd import HipGPEngine

    return HipGPEngine(dtype, **kw)


def _theta(d, y, kernel="Matern52", noise=1e-3, ard=False, variance=1.3):
    ls = 0.25 * np.sqrt(d) * (np.linspace(0.8, 1.3, d) if ard else np.ones(1))
    return gpr.Theta(kernel, ls, variance, noise, float(y.mean()))


def _fit(eng, X, y, th, grad=False):
    eng.set_data(X, y)
    return eng.fit_eval(th.kernel, th.lengthscales, th.variance, th.noise, th.mean_c, want_grad=grad)


def _rel(a, b):
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(1e-300, np.max(np.abs(b))))


@pytest.mark.parametrize("k", [1, 7, 64])
@pytest.mark.parametrize("n,d,ard", [(130, 2, False), (300, 5, True), (2048 - 64, 12, False), (4096 - 64, 6, False)])
def test_append_fp64_matches_the_oracles_from_scratch_posterior(n, d, ard, k):
    from pygpso_amd import _lib as L

    X, y = synthetic_problem(n + k, d, seed=3)
    th = _theta(d, y, ard=ard)
    post = gpr.posterior(th, X, y)  # the N + k points, from scratch
    f_ref, _ = gpr.nlml_and_grad(th, X, y)
    eng = _engine()
    _fit(eng, X[:n], y[:n], th, grad=(k == 7))  # (an evaluation with gradient leaves the same factor)
    f, in_place = eng.append(X[n:], y[n:])
    # (more than 32 points beside a factor below 4096 rows: the library refits instead -- the k x k corner is one workgroup's)
    assert in_place == (not (k > 32 and eng.padded_n < 4096)), eng.last_message()
    assert eng.n == n + k
    assert abs(f - f_ref) <= 1e-9 * abs(f_ref)
    Linv_ref = np.linalg.inv(post.L)
    assert _rel(eng.get_matrix(L.MAT_CHOL), post.L) < 1e-9
    assert _rel(eng.get_matrix(L.MAT_LINV), Linv_ref) < 1e-8
    assert _rel(eng.get_vector(L.VEC_ALPHA), post.alpha) < 1e-8
    assert _rel(eng.get_vector(L.VEC_WHITE), Linv_ref @ (y - th.mean_c)) < 1e-8
    Xs = synthetic_leaves(1500, d)
    mean, var = eng.predict(Xs)
    mean_ref, var_ref = gpr.predict_y(post, Xs)
    scale = max(1.0, float(np.max(np.abs(y))))
    assert np.max(np.abs(mean - mean_ref)) <= 1e-9 * scale
    assert np.max(np.abs(var - var_ref)) <= 1e-9 * th.variance
    idx, mu, vv, ucb = eng.best_ucb(Xs, VS)
    i_ref = gpr.best_ucb(post, Xs)[0]
    assert int(idx[0]) == i_ref
    # ... and a fit from scratch on the same context afterwards sees the same N + k points
    f2, _ = eng.fit_eval(th.kernel, th.lengthscales, th.variance, th.noise, th.mean_c, want_grad=False)
    assert abs(f2 - f_ref) <= 1e-9 * abs(f_ref)


@pytest.mark.parametrize("kernel", ["Matern32", "Matern12", "SquaredExponential"])
def test_append_fp64_other_kernels(kernel):
    n, d, k = 200, 3, 5
    tol = 1e-5 if kernel == "Matern12" else 1e-9
    X, y = synthetic_problem(n + k, d, seed=5)
    th = _theta(d, y, kernel=kernel)
    post = gpr.posterior(th, X, y)
    f_ref, _ = gpr.nlml_and_grad(th, X, y)
    eng = _engine()
    _fit(eng, X[:n], y[:n], th)
    f, in_place = eng.append(X[n:], y[n:])
    assert in_place
    assert abs(f - f_ref) <= tol * abs(f_ref)
    Xs = synthetic_leaves(500, d)
    mean, var = eng.predict(Xs)
    mean_ref, var_ref = gpr.predict_y(post, Xs)
    assert np.max(np.abs(mean - mean_ref)) <= tol * 10 * max(1.0, float(np.max(np.abs(y))))
    assert np.max(np.abs(var - var_ref)) <= tol * 10 * th.variance


def test_a_chain_of_twenty_appends_equals_one_fit_fp64():
    n0, d = 150, 4
    ks = [1, 2, 7, 3, 1, 5, 1, 1, 4, 6, 2, 1, 7, 1, 3, 2, 1, 1, 5, 4]
    n1 = n0 + sum(ks)
    X, y = synthetic_problem(n1, d, seed=7)
    th = _theta(d, y)
    eng = _engine()
    _fit(eng, X[:n0], y[:n0], th)
    lo = n0
    for k in ks:
        f, in_place = eng.append(X[lo:lo + k], y[lo:lo + k])
        assert in_place
        lo += k
    post = gpr.posterior(th, X, y)
    f_ref, _ = gpr.nlml_and_grad(th, X, y)
    assert abs(f - f_ref) <= 1e-9 * abs(f_ref)
    Xs = synthetic_leaves(800, d)
    mean, var = eng.predict(Xs)
    mean_ref, var_ref = gpr.predict_y(post, Xs)
    assert np.max(np.abs(mean - mean_ref)) <= 1e-9 * max(1.0, float(np.max(np.abs(y))))
    assert np.max(np.abs(var - var_ref)) <= 1e-9 * th.variance
    # the same chain on a second context gives the same bits (deterministic reductions)
    eng2 = _engine()
    _fit(eng2, X[:n0], y[:n0], th)
    lo = n0
    for k in ks:
        eng2.append(X[lo:lo + k], y[lo:lo + k])
        lo += k
    m2, v2 = eng2.predict(Xs)
    assert np.array_equal(mean, m2) and np.array_equal(var, v2)


def test_pad_crossing_small_problems_and_wide_blocks_refit_and_say_so():
    d = 3
    X, y = synthetic_problem(400, d, seed=9)
    th = _theta(d, y)
    for n, k, why in [(126, 5, "padded size"), (40, 6, "one-launch"), (200, 70, "more than 64")]:
        eng = _engine()
        _fit(eng, X[:n], y[:n], th)
        f, in_place = eng.append(X[n:n + k], y[n:n + k])
        assert not in_place and why in eng.last_message(), eng.last_message()
        post = gpr.posterior(th, X[:n + k], y[:n + k])
        f_ref, _ = gpr.nlml_and_grad(th, X[:n + k], y[:n + k])
        assert abs(f - f_ref) <= 1e-9 * abs(f_ref)
        Xs = synthetic_leaves(300, d)
        mean, var = eng.predict(Xs)
        mean_ref, var_ref = gpr.predict_y(post, Xs)
        assert np.max(np.abs(mean - mean_ref)) <= 1e-9 * max(1.0, float(np.max(np.abs(y))))
        assert np.max(np.abs(var - var_ref)) <= 1e-9 * th.variance
    # float-predict contexts pad to 256: the crossing is there
    eng = _engine("mixed")
    _fit(eng, X[:250], y[:250], th)
    _, in_place = eng.append(X[250:257], y[250:257])
    assert not in_place and "padded size" in eng.last_message()


def test_a_block_that_is_not_positive_definite_leaves_the_posterior_alone():
    n, d = 300, 3
    X, y = synthetic_problem(n, d, seed=11)
    th = _theta(d, y, noise=1e-6)
    eng = _engine()
    _fit(eng, X, y, th)
    Xs = synthetic_leaves(400, d)
    before = eng.predict(Xs)
    bad = np.vstack([X[:2], np.full((1, d), np.nan)])  # a NaN input: the Schur complement cannot be factorised
    with pytest.raises(np.linalg.LinAlgError) as err:
        eng.append(bad, np.zeros(3))
    assert "pivot" in str(err.value)
    assert eng.n == n
    after = eng.predict(Xs)
    assert np.array_equal(before[0], after[0]) and np.array_equal(before[1], after[1])
    # state errors: no posterior yet
    eng2 = _engine()
    eng2.set_data(X, y)
    from pygpso_amd import _lib as L

    with pytest.raises(L.GpsoHipError):
        eng2.append(X[:1], y[:1])


@pytest.mark.parametrize("math", ["f16x3", "bf16x6", "native"])
@pytest.mark.parametrize("dtype", ["mixed", "float32"])
def test_append_in_float_contexts_repacks_only_the_new_tile_rows(dtype, math):
    """The 16-bit pieces / the packed f32 tiles of the rows an append wrote are packed in place; everything a from-scratch
    fit of the same context type would be held to is held here: the float bounds of tests/test_gpu_parity.py."""
    n, d, k = 2048 - 40, 12, 7
    X, y = synthetic_problem(n + 3 * k, d, seed=13)
    th = gpr.Theta("Matern52", np.array([0.25 * np.sqrt(d)]), 1.0, 1e-3, float(y.mean()))
    Xs = synthetic_leaves(4096, d)
    eng = _engine(dtype, predict_math=math)
    _fit(eng, X[:n], y[:n], th)
    eng.predict(Xs[:256])  # (the self-test has ruled and the pieces are built before the first append)
    lo = n
    for _ in range(3):
        f, in_place = eng.append(X[lo:lo + k], y[lo:lo + k])
        assert in_place, eng.last_message()
        lo += k
    post = gpr.posterior(th, X, y)
    f_ref, _ = gpr.nlml_and_grad(th, X, y)
    assert abs(f - f_ref) <= (2e-5 if dtype == "float32" else 1e-9) * abs(f_ref)
    mean, var = eng.predict(Xs)
    assert eng.precision_info()["predict_math"] == math
    mean_ref, var_ref = gpr.predict_y(post, Xs)
    em = float(np.max(np.abs(mean - mean_ref)) / np.max(np.abs(y)))
    ev = float(np.max(np.abs(var - var_ref)) / th.variance)
    print(f"append {dtype}/{math}: |d mean| {em:.2e} max|y|, |d var| {ev:.2e} sigma^2")
    assert em <= 1.8e-4 and ev <= 1.3e-5  # FLOAT_BOUNDS["C3"] of tests/test_gpu_parity.py
    # against a from-scratch fit of the same context type on the same points: the same class of error
    ref = _engine(dtype, predict_math=math)
    _fit(ref, X, y, th)
    m2, v2 = ref.predict(Xs)
    em2 = float(np.max(np.abs(m2 - mean_ref)) / np.max(np.abs(y)))
    ev2 = float(np.max(np.abs(v2 - var_ref)) / th.variance)
    assert em <= 4 * em2 + 1e-6 and ev <= 4 * ev2 + 1e-7, (em, em2, ev, ev2)


def test_append_when_the_fp16_scale_crosses_a_power_of_two():
    """A new point almost on top of an old one beside SPARSE old points (D = 12, short lengthscale: max |L^-1| ~ 1 before,
    1 / sqrt(2 noise) ~ 70 after): the largest entry jumps six powers of two and the fp16 pieces of EVERY row are repacked
    with the new scale (decided on the device).  (Round 5's recipe -- D = 2, 500 dense points -- did not cross: among dense
    points every conditional variance is already ~ noise.  The crossing is now asserted: gpso_posterior_dirty_ranges reports
    the whole span only when the scale moved.)"""
    n, d = 300, 12
    X, y = synthetic_problem(n + 1, d, seed=17)
    X[n] = X[3] + 1e-7
    y[n] = y[3]
    th = gpr.Theta("Matern52", np.array([0.2]), 1.0, 1e-4, float(y.mean()))
    eng = _engine("mixed", predict_math="f16x3", precision_check=False)
    _fit(eng, X[:n], y[:n], th)
    Xs = synthetic_leaves(2000, d)
    eng.predict(Xs[:256])
    eng.posterior_mark_synced()
    assert eng.posterior_dirty_ranges() == []
    _, in_place = eng.append(X[n:], y[n:])
    assert in_place
    assert len(eng.posterior_dirty_ranges()) == 1  # the scale moved: nothing a peer holds is still valid
    post = gpr.posterior(th, X, y)
    mean, var = eng.predict(Xs)
    mean_ref, var_ref = gpr.predict_y(post, Xs)
    assert np.max(np.abs(mean - mean_ref)) <= 6e-4 * np.max(np.abs(y))
    assert np.max(np.abs(var - var_ref)) <= 1e-4 * th.variance


def test_append_data_on_the_model_and_refit_every_on_the_surrogate():
    from pygpso_amd import GPRSurrogate
    from pygpso_amd.kernels import Constant, Matern52

    n, d = 160, 2
    X, y = synthetic_problem(n + 12, d, seed=19)
    surr = GPRSurrogate(gp_kernel=Matern52(lengthscales=0.25, variance=1.0), gp_meanf=Constant(0.0), refit_every=3)
    surr.append(X[:n], y[:n])
    surr.gp_update()  # update 0: hyper-parameters optimised
    model = surr.gpflow_model
    evals0 = model.num_loss_evals
    theta0 = model.parameter_dict()
    for step in range(1, 4):
        lo = n + 4 * (step - 1)
        surr.append(X[lo:lo + 4], y[lo:lo + 4])
        surr.gp_update()
        if step < 3:  # appended at the kept hyper-parameters: no loss evaluation
            assert model.num_loss_evals == evals0
            assert all(np.array_equal(theta0[k], v) for k, v in model.parameter_dict().items())
            th = gpr.Theta("Matern52", np.atleast_1d(theta0[".kernel.lengthscales"]), float(theta0[".kernel.variance"]),
                           float(theta0[".likelihood.variance"]), float(theta0[".mean_function.c"]))
            post = gpr.posterior(th, X[:lo + 4], y[:lo + 4])
            Xs = synthetic_leaves(200, d)
            mean, var = model.predict_y(Xs)
            mean_ref, var_ref = gpr.predict_y(post, Xs)
            assert np.max(np.abs(mean[:, 0] - mean_ref)) <= 1e-8 * max(1.0, float(np.max(np.abs(y))))
            assert np.max(np.abs(var[:, 0] - var_ref)) <= 1e-8 * th.variance
        else:  # the third update re-optimises
            assert model.num_loss_evals > evals0


@pytest.mark.parametrize("dtype", ["mixed", "float32"])
def test_append_behind_an_evaluation_with_gradient_and_a_wide_block(dtype):
    """(a) An evaluation WITH gradient defers the packing of the 16-bit pieces: an append behind it must leave the later
    packing the right maximum (the fp16 scale) and the right N.  (b) k = 40 runs the 64-wide instantiation of the passes in a
    float context (N_pad = 4096: below that a block this wide is refitted)."""
    n, d, k = 4096 - 47, 6, 40
    X, y = synthetic_problem(n + k + 7, d, seed=23)
    th = gpr.Theta("Matern52", np.array([0.25 * np.sqrt(d)]), 1.0, 1e-3, float(y.mean()))
    eng = _engine(dtype)
    _fit(eng, X[:n], y[:n], th, grad=True)          # pieces pending
    f1, in_place = eng.append(X[n:n + 7], y[n:n + 7])
    assert in_place, eng.last_message()
    f2, in_place = eng.append(X[n + 7:], y[n + 7:])  # k = 40 -> KP = 64
    assert in_place, eng.last_message()
    post = gpr.posterior(th, X, y)
    assert abs(f2 - post.nlml) <= (2e-5 if dtype == "float32" else 1e-9) * abs(post.nlml)
    Xs = synthetic_leaves(2048, d)
    mean, var = eng.predict(Xs)
    mean_ref, var_ref = gpr.predict_y(post, Xs)
    em = float(np.max(np.abs(mean - mean_ref)) / np.max(np.abs(y)))
    ev = float(np.max(np.abs(var - var_ref)) / th.variance)
    print(f"append behind a gradient, k = 7 + 40, {dtype}: |d mean| {em:.2e} max|y|, |d var| {ev:.2e} sigma^2")
    assert em <= 3.3e-4 and ev <= 2.4e-5  # the C4-class float bounds of tests/test_gpu_parity.py (N = 4096 .. 8192)


def test_an_appended_posterior_travels_like_a_fitted_one():
    """The extended posterior is handed to a second context as ONE span copy (what gpso_broadcast_posterior moves): the
    receiver adopts N + k points and predicts the sender's bits; fingerprints of two contexts that ran the same fit + append
    agree (a replicating group appends on every rank)."""
    import torch

    from tests.test_gpu_distributed import _DeviceBytes

    n, d, k = 600, 4, 5
    X, y = synthetic_problem(n + k, d, seed=29)
    th = gpr.Theta("Matern52", np.array([0.5]), 1.0, 1e-3, float(y.mean()))
    Xs = synthetic_leaves(3000, d)
    for dtype in ("float64", "mixed"):
        a, b, c = _engine(dtype), _engine(dtype), _engine(dtype)
        for e in (a, c):
            _fit(e, X[:n], y[:n], th)
            e.append(X[n:], y[n:])
        assert a.posterior_hash() == c.posterior_hash()
        ptr, off, nb = a.posterior_span()
        b.alloc_posterior(a.n, a.d)
        dst = b.posterior_span_at(off, nb)
        torch.as_tensor(_DeviceBytes(dst, nb), device="cuda").copy_(torch.as_tensor(_DeviceBytes(ptr, nb), device="cuda"))
        torch.cuda.synchronize()
        b.adopt_posterior()
        assert b.n == n + k
        ra, rb = a.best_ucb(Xs, VS), b.best_ucb(Xs, VS)
        assert all(np.array_equal(u, v) for u, v in zip(ra, rb))
        ma, mb = a.predict(Xs), b.predict(Xs)
        assert np.array_equal(ma[0], mb[0]) and np.array_equal(ma[1], mb[1])


# ---- round 6: the append at the sizes it was built for (BASELINE configs C4 / C5) -------------------------------------------
def _append_at_size(family, n1, d, k, dtype, post_nlml, leaves_sub, mean_ref, var_ref, X, y, th, post=None):
    """Fit the first n1 - k points, append the last k, compare with the ORACLE's from-scratch posterior of all n1 (one CPU
    factorisation shared with tests/test_gpu_parity.py's config tests).  Float contexts: FLOAT_BOUNDS[family] unchanged,
    NLML 2e-5; float64: NLML 1e-9, predictions 1e-8 (the bound of the from-scratch fit at N = 4600), the new rows of L 1e-9."""
    from pygpso_amd import _lib as L
    from tests.test_gpu_parity import FLOAT_BOUNDS

    n = n1 - k
    eng = _engine(dtype)
    _fit(eng, X[:n], y[:n], th)
    if dtype != "float64":
        eng.predict(leaves_sub[:64])  # (the self-test has ruled and the 16-bit pieces exist: the append repacks them in place)
    f, in_place = eng.append(X[n:], y[n:])
    assert in_place, eng.last_message()
    assert eng.n == n1
    tol_f = 2e-5 if dtype == "float32" else 1e-9
    assert abs(f - post_nlml) <= tol_f * abs(post_nlml), (f, post_nlml)
    mean, var = eng.predict(leaves_sub)
    ys = max(1.0, float(np.max(np.abs(y))))
    em, ev = float(np.max(np.abs(mean - mean_ref)) / ys), float(np.max(np.abs(var - var_ref)) / th.variance)
    print(f"append at {family} (N = {n} + {k}, {dtype}): nlml {abs(f - post_nlml) / abs(post_nlml):.1e}, |d mean| {em:.2e} max|y|, "
          f"|d var| {ev:.2e} sigma^2")
    if dtype == "float64":
        assert em <= 1e-8 and ev <= 1e-8
        if post is not None:  # the rows the append wrote, against the oracle's factor
            chol = eng.get_matrix(L.MAT_CHOL)
            assert _rel(chol[n:], post.L[n:]) < 1e-9
            assert _rel(eng.get_vector(L.VEC_ALPHA), post.alpha) < 1e-6  # (a forward error: cond(K_y) ~ 1e6 at this size)
    else:
        bm, bv = FLOAT_BOUNDS[family]
        assert em <= bm and ev <= bv
    return eng


@pytest.mark.parametrize("k", [1, 7, 40])
@pytest.mark.parametrize("dtype", ["float32", "mixed", "float64"])
def test_append_at_config_C4_size_matches_the_oracles_from_scratch_posterior(dtype, k):
    from tests.test_gpu_parity import _c4_posterior, _problem

    n1, d = 8192, 20
    X, y, th = _problem(n1, d, variance=1.0)
    post = _c4_posterior(th, X, y)
    Xs = synthetic_leaves(32768, d)
    sub = np.sort(np.random.default_rng(31).choice(Xs.shape[0], 256, replace=False))
    mean_ref, var_ref = gpr.predict_y(post, Xs[sub])
    _append_at_size("C4", n1, d, k, dtype, post.nlml, Xs[sub], mean_ref, var_ref, X, y, th, post)


@pytest.mark.parametrize("k", [1, 7, 40])
@pytest.mark.parametrize("dtype", ["float32", "mixed"])
def test_append_at_config_C5_size_matches_the_oracles_from_scratch_posterior(dtype, k):
    """N = 16 384 - k: the passes run with the nq = 8 chunking and 256 tiles per block column.  The oracle is the CPU potrf
    at 16 384 the C5 tests of tests/test_gpu_parity.py already pay for (noise 1e-3: bench.py --workload c5's)."""
    from tests.test_gpu_parity import _c5_reference

    ref = _c5_reference(1e-3)
    X, y, th = ref["X"], ref["y"], ref["th"]
    leaves = synthetic_leaves(131072, 40).astype(np.float32)[ref["sub"]].astype(np.float64)
    _append_at_size("C5", 16384, 40, k, dtype, ref["nlml"], leaves, ref["mean_ref"], ref["var_ref"], X, y, th)


def test_an_appended_C4_posterior_through_the_eight_shard_replay():
    """The appended C4 posterior handed to a second context as one span copy, scored as 8 shards on the two contexts
    (gpso_shard_winners x 8 + gpso_fold_winners: the group call minus the ncclAllGather) = ONE gpso_best_ucb over the
    batch, bit for bit -- and the winner is the oracle's (float rule)."""
    from tests.helpers import winner_is_the_oracles
    from tests.test_gpu_distributed import _replay_at_size
    from tests.test_gpu_parity import _c4_posterior, _problem

    n1, d, k, m = 8192, 20, 7, 32768
    X, y, th = _problem(n1, d, variance=1.0)
    post = _c4_posterior(th, X, y)
    root = _engine("float32")
    _fit(root, X[:n1 - k], y[:n1 - k], th)
    Xs = synthetic_leaves(m, d).astype(np.float32)
    root.predict(Xs[:256])
    _, in_place = root.append(X[n1 - k:], y[n1 - k:])
    assert in_place and root.n == n1
    segs = {"one": None, "ragged": np.array([0, 1000, 1000, m // 8 + 5, m // 2, m - 1, m], dtype=np.int64)}
    whole, _ = _replay_at_size(root, Xs, segs)
    mean_ref, var_ref = gpr.predict_y(post, Xs.astype(np.float64))
    winner_is_the_oracles(whole["one"][0][0], mean_ref + VS * var_ref, 2e-5, "test_gpu_append")


@pytest.mark.parametrize("k", [33, 48, 64])
def test_wide_blocks_in_a_float32_context(k):
    """k = 33 ... 64 (the 64-wide instantiation of the passes and of the one-workgroup corner) in a float32 context at
    N_pad = 4096, against the oracle: the float bounds of the from-scratch fit at this size (C4 class)."""
    n1, d = 4096, 6
    X, y = synthetic_problem(n1, d, seed=37)
    th = gpr.Theta("Matern52", np.array([0.25 * np.sqrt(d)]), 1.0, 1e-3, float(y.mean()))
    post = gpr.posterior(th, X, y)
    Xs = synthetic_leaves(1024, d)
    mean_ref, var_ref = gpr.predict_y(post, Xs)
    _append_at_size("C4", n1, d, k, "float32", post.nlml, Xs, mean_ref, var_ref, X, y, th)


# ---- round 6: failures leave ONE consistent state ---------------------------------------------------------------
def test_a_failed_refit_path_append_restores_the_posterior_of_the_first_points():
    """k > 64 takes the library's refit path (gpso_set_data + gpso_fit_eval on the N + k points).  When THAT fit fails the
    header's promise still holds: the posterior of the first N points is resident again, N is N, predictions are the bits
    of before."""
    n, d = 300, 3
    X, y = synthetic_problem(n, d, seed=41)
    th = _theta(d, y, noise=1e-6)
    eng = _engine()
    _fit(eng, X, y, th)
    Xs = synthetic_leaves(400, d)
    before = eng.predict(Xs)
    bad = np.vstack([synthetic_problem(69, d, seed=43)[0], np.full((1, d), np.nan)])  # 70 rows: the refit path; a NaN input
    with pytest.raises(np.linalg.LinAlgError) as err:
        eng.append(bad, np.zeros(70))
    assert "resident again" in str(err.value)
    assert eng.n == n
    after = eng.predict(Xs)
    assert np.array_equal(before[0], after[0]) and np.array_equal(before[1], after[1])
    f, in_place = eng.append(X[:1] + 0.01, y[:1])  # ... and the context appends as before
    assert in_place and eng.n == n + 1


def test_a_failed_append_leaves_the_model_consistent_and_the_surrogate_reoptimises():
    from pygpso_amd import GPRSurrogate
    from pygpso_amd.kernels import Constant, Matern52
    from pygpso_amd.model import HipGPR

    n, d = 200, 2
    X, y = synthetic_problem(n + 4, d, seed=47)
    model = HipGPR(data=(X[:n], y[:n, None]), kernel=Matern52(lengthscales=0.3, variance=1.0), mean_function=Constant(0.0),
                   noise_variance=1e-4)
    model.predict_y(X[:3])
    bad = np.vstack([X[n:n + 2], np.full((1, d), np.nan)])
    with pytest.raises(np.linalg.LinAlgError):
        model.append_data(bad, np.zeros(3))
    # the model claims the N + 3 rows as data and NO resident posterior (the advisor's case: it claimed a posterior of
    # N + 3 points while the device held N)
    assert model.data[0].shape[0] == n + 3 and not model._resident and model._device_theta is None
    model.data = (X[:n], y[:n, None])
    mean, _ = model.predict_y(X[:3])
    assert np.all(np.isfinite(mean))
    # the surrogate: an update whose append fails re-optimises on all points instead
    surr = GPRSurrogate(gp_kernel=Matern52(lengthscales=0.25, variance=1.0), gp_meanf=Constant(0.0), refit_every=4)
    surr.append(X[:n], y[:n])
    surr.gp_update()
    m = surr.gpflow_model
    evals0 = m.num_loss_evals
    real_append = m.engine.append

    def failing(*a, **kw):  # (a NaN row would also poison the re-optimisation: the failure itself is forced here, the
        raise np.linalg.LinAlgError("forced: the appended block is not positive definite")  # device's is tested above)

    m.engine.append = failing
    surr.append(X[n:n + 4], y[n:n + 4])
    surr.gp_update()
    m.engine.append = real_append
    assert m.num_loss_evals > evals0  # re-optimised on all N + 4 points
    assert m.data[0].shape[0] == n + 4 == surr.num_evaluated and m.engine.n == n + 4
    th = m.parameter_dict()
    post = gpr.posterior(gpr.Theta("Matern52", np.atleast_1d(th[".kernel.lengthscales"]), float(th[".kernel.variance"]),
                                   float(th[".likelihood.variance"]), float(th[".mean_function.c"])), X, y)
    mean, var = m.predict_y(X[:5])
    mean_ref, var_ref = gpr.predict_y(post, X[:5])
    assert np.max(np.abs(mean[:, 0] - mean_ref)) <= 1e-8 and np.max(np.abs(var[:, 0] - var_ref)) <= 1e-8

"""GPU checks of the multi-GPU plumbing that can run on ONE device: the posterior hand-off protocol on
real device memory (what a broadcast does, spelled out with device-to-device copies), and the RCCL
group calls of the C-ABI in a world of size 1 -- per-rank style and the thread-driven
``HipGPEngineGroup`` behind ``GPRSurrogate(devices=[...])``.  World sizes > 1 on the DEVICE: the two halves of the
sharded calls (gpso_shard_winners / gpso_shard_winners_grow / gpso_fold_winners: the group calls' own kernels and
index arithmetic, the all-gather replaced by a concatenation) replay groups of 2, 3 and 8 ranks on two contexts of
this one GPU and must reproduce the single-context call bit for bit.  (RCCL itself with more than one rank: the
driver's scaling run; the host-side choreography: the world-2 ``gloo`` tests of tests/test_distributed_cpu.py.)"""
import numpy as np
import pytest

from oracle import gpr, tree
from tests.helpers import load_goldens, rotated_peaks, synthetic_leaves, synthetic_problem

pytestmark = pytest.mark.gpu
VS = gpr.VARSIGMA_DEFAULT
G = load_goldens()


class _DeviceBytes:
    """Expose a raw device allocation to torch (TEST plumbing) through ``__cuda_array_interface__``."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False),
                                         "version": 2, "strides": None}


def _posterior_tensors(engine):
    import torch

    return [torch.as_tensor(_DeviceBytes(p, nb), device=torch.device("cuda", engine.device))
            for p, nb in engine.posterior_buffers()]


def _fitted(dtype, n=300, **opts):
    from pygpso_amd import HipGPEngine

    X, y = synthetic_problem(n, 5, seed=0)
    eng = HipGPEngine(dtype, **opts)
    eng.set_data(X, y)
    eng.fit_eval("Matern52", [0.5], 1.0, 1e-3, float(y.mean()), want_grad=False)
    return eng


def _handoff(src, dst):
    import torch

    dst.alloc_posterior(src.n, src.d)
    a, b = _posterior_tensors(src), _posterior_tensors(dst)
    assert [t.numel() for t in a] == [t.numel() for t in b]
    for s_, t_ in zip(a, b):
        t_.copy_(s_)
    torch.cuda.synchronize()
    dst.adopt_posterior()
    return len(a)


def _handoff_span(src, dst):
    """What gpso_broadcast_posterior does since round 4, spelled out with ONE device-to-device copy: the contiguous
    range of the sender's posterior arena the posterior uses, into the same offset of the receiver's arena."""
    import torch

    ptr, off, nb = src.posterior_span()
    dst.alloc_posterior(src.n, src.d)
    dev = torch.device("cuda", src.device)
    a = torch.as_tensor(_DeviceBytes(ptr, nb), device=dev)
    b = torch.as_tensor(_DeviceBytes(dst.posterior_span_at(off, nb), nb), device=dev)
    b.copy_(a)
    torch.cuda.synchronize()
    dst.adopt_posterior()
    return nb


@pytest.mark.parametrize("dtype,math,n", [("float64", None, 300), ("float32", "native", 512), ("float32", "auto", 512),
                                          ("mixed", "f16x3", 512), ("float32", "bf16x6", 768), ("mixed", "bf16x3", 256),
                                          ("float32", "auto", 300), ("mixed", "auto", 100)])
def test_posterior_span_is_one_contiguous_range_and_enough_to_predict(dtype, math, n):
    """Round 4: every posterior buffer is a slice of ONE allocation; the range the resident posterior uses is
    contiguous whichever predict math it runs, sits at the same offset on every context, and a receiver that got
    nothing but that range predicts bit-identically.  A split-math posterior travels WITHOUT the packed f32 L^-1
    (2 x 2 bytes per entry instead of + 2 more): asking the receiver for the f32 kernel is refused, not garbage."""
    from pygpso_amd import HipGPEngine
    from pygpso_amd import _lib as L

    opts = {} if math is None else {"predict_math": math}
    src = _fitted(dtype, n=n, **opts)
    dst = HipGPEngine(dtype, **opts)
    Xs = synthetic_leaves(1500, 5)
    exp = src.predict(Xs)
    split = src.precision_info()["predict_math"] != "native"
    nb = _handoff_span(src, dst)
    bufs = dict(zip(("hyper", "linv_p", "xs", "xs_p", "xnorm", "alpha", "linv_b"), src.posterior_buffers()))
    ptr, off, _ = src.posterior_span()
    # the slices the span must cover lie inside it, in one piece
    need = ["hyper", "xs", "xs_p", "xnorm", "alpha"] + (["linv_b"] if split else ["linv_p"])
    for name in need:
        p, b = bufs[name]
        used = b if name != "linv_b" else 256 + {"f16x3": 2, "bf16x3": 2, "bf16x6": 3}[src.precision_info()["predict_math"]] * src.padded_n ** 2 * 2
        assert ptr <= p and p + used <= ptr + nb, name
    small = sum(bufs[k][1] for k in ("hyper", "xs", "xs_p", "xnorm", "alpha"))
    big = (256 + (3 if src.precision_info()["predict_math"] == "bf16x6" else 2) * src.padded_n ** 2 * 2) if split else bufs["linv_p"][1]
    assert big + small <= nb <= big + small + 6 * 256  # (slices are 256-byte aligned)
    assert dst.posterior_span()[1:] == (off, nb)  # same layout on the receiver
    got = dst.predict(Xs)
    assert np.array_equal(exp[0], got[0]) and np.array_equal(exp[1], got[1])
    assert all(np.array_equal(a, b) for a, b in zip(src.best_ucb(Xs, VS), dst.best_ucb(Xs, VS)))
    if split:
        dst.set_predict_math("native")
        with pytest.raises(L.GpsoHipError, match="split pieces"):
            dst.predict(Xs)
    with pytest.raises(ValueError):
        dst.posterior_span_at(off + 1, nb)


def test_posterior_handoff_carries_the_contraction_choice():
    """A posterior whose self-test kept float generation with the f32 contraction (the second rung of GPSO_GEN_AUTO:
    tests/test_gpu_parity.py::test_generation_choice_*) says so in its hyper block: the receiver of its span predicts
    with the same arithmetic -- the same bits -- not with its own default."""
    from pygpso_amd import HipGPEngine

    n, d = 512, 4
    X, y = synthetic_problem(n, d, seed=0)
    src = HipGPEngine("float32")
    src.set_data(X, y)
    src.fit_eval("Matern52", [0.25 * np.sqrt(d)], 1.0, 1e-3, float(y.mean()), want_grad=False)
    Xs = synthetic_leaves(900, d, seed=2)
    exp = src.predict(Xs)
    forced = HipGPEngine("float32", generation="float32")
    forced.set_contraction("f16")
    forced.set_data(X, y)
    forced.fit_eval("Matern52", [0.25 * np.sqrt(d)], 1.0, 1e-3, float(y.mean()), want_grad=False)
    assert not np.array_equal(forced.predict(Xs)[1], exp[1])  # (the sender is NOT on the default contraction)
    dst = HipGPEngine("float32")
    _handoff_span(src, dst)
    got = dst.predict(Xs)
    assert np.array_equal(exp[0], got[0]) and np.array_equal(exp[1], got[1])


@pytest.mark.parametrize("dtype,math", [("float64", None), ("float32", "auto"), ("mixed", "bf16x6"), ("float32", "native")])
def test_posterior_fingerprints_agree_between_replicated_fits(dtype, math):
    """gpso_posterior_hash: two contexts that ran the same fit hold the same fingerprint (the fit is bit-deterministic:
    what a group with posterior="replicate" relies on), a receiver of the posterior holds it too, another theta does not."""
    from pygpso_amd import HipGPEngine

    opts = {} if math is None else {"predict_math": math}
    a, b = _fitted(dtype, n=512, **opts), _fitted(dtype, n=512, **opts)
    h = a.posterior_hash()
    assert h == b.posterior_hash() == a.posterior_hash()
    c = HipGPEngine(dtype, **opts)
    _handoff_span(a, c)
    assert c.posterior_hash() == h
    X, y = synthetic_problem(512, 5, seed=0)
    b.fit_eval("Matern52", [0.5], 1.0, 1.0000001e-3, float(y.mean()), want_grad=False)
    assert b.posterior_hash() != h


def test_replicating_group_of_one_device_and_its_fingerprint_check():
    """``HipGPEngineGroup(posterior="replicate")``: every device fits its own copy (no broadcast); the ranks'
    fingerprints are compared before the first predict-type call on a new posterior."""
    from pygpso_amd import _lib as L
    from pygpso_amd.distributed import HipGPEngineGroup

    X, y = synthetic_problem(300, 5, seed=0)
    plain = _fitted("float64")
    grp = HipGPEngineGroup("float64", devices=[0], posterior="replicate")
    grp.set_data(X, y)
    grp.fit_eval("Matern52", [0.5], 1.0, 1e-3, float(y.mean()), want_grad=False)
    Xs = synthetic_leaves(2500, 5)
    assert all(np.array_equal(a, b) for a, b in zip(grp.best_ucb(Xs, VS), plain.best_ucb(Xs, VS)))
    # two "devices" that disagree (world 2 faked on the host side of the group: the check itself)
    grp.world = 2
    grp.engines = [grp.engines[0], _fitted("float64", n=301)]
    grp._stale = True
    with pytest.raises(L.GpsoHipError, match="fingerprints"):
        grp._sync_posterior()
    grp.world = 1
    grp.engines = grp.engines[:1]
    grp.close()


def test_posterior_handoff_carries_the_bf16_pieces():
    from pygpso_amd import HipGPEngine

    src = _fitted("float32", n=512, predict_math="bf16x6")
    dst = HipGPEngine("float32", predict_math="bf16x6")
    assert _handoff(src, dst) == 7
    Xs = synthetic_leaves(1500, 5)
    assert all(np.array_equal(p, q) for p, q in zip(src.predict(Xs), dst.predict(Xs)))


@pytest.mark.parametrize("dtype", ["float64", "float32", "mixed"])
def test_posterior_handoff_between_two_contexts(dtype):
    """The receiver predicts bit-identically to the context that fitted."""
    from pygpso_amd import HipGPEngine

    src = _fitted(dtype)
    dst = HipGPEngine(dtype)
    # (N = 300: float-predict contexts pad to 512 = a multiple of the split kernels' row block, so the fp16 pieces exist
    # and travel as the seventh buffer; a float64 context pads to 384 and has six)
    assert _handoff(src, dst) == (6 if dtype == "float64" else 7)
    assert src.padded_n == (384 if dtype == "float64" else 512)
    if dtype != "float64":
        assert src.precision_info()["predict_math"] == "f16x3"
    Xs = synthetic_leaves(2000, 5)
    m1, v1 = src.predict(Xs)
    m2, v2 = dst.predict(Xs)
    assert np.array_equal(m1, m2) and np.array_equal(v1, v2)
    assert all(np.array_equal(p, q) for p, q in zip(src.best_ucb(Xs, VS), dst.best_ucb(Xs, VS)))


def test_the_packed_posterior_holds_lower_tiles_only():
    """L^-1 travels as its lower 16x16 tiles (N_pad/16 (N_pad/16 + 1) / 2 of them): about half of the
    square round 1 broadcast."""
    eng = _fitted("float32", n=2048)
    sizes = dict(zip(("hyper", "linv_p", "xs", "xs_p", "xnorm", "alpha"), (nb for _, nb in eng.posterior_buffers())))
    t = eng.padded_n // 16
    assert sizes["linv_p"] == t * (t + 1) // 2 * 256 * 4 < 0.51 * eng.padded_n ** 2 * 4 + 16 * eng.padded_n * 4


def test_group_calls_in_a_world_of_one():
    """gpso_comm_init / gpso_broadcast_posterior / gpso_best_ucb_sharded / gpso_best_ucb_grow_sharded on
    a real RCCL communicator of one rank: same results, bit for bit, as the single-GPU calls."""
    from pygpso_amd import distributed as D
    from pygpso_amd import _lib as L

    eng = _fitted("float32")
    with pytest.raises(L.GpsoHipError):  # group calls need a group
        eng.best_ucb_grow_sharded(np.array([[(0.0, 1.0)] * 5]), 3, VS)
    eng.comm_init(0, 1, D.exchange_unique_id(0, 1))
    try:
        D.broadcast_posterior(eng, src=0)  # RCCL broadcast straight on the library's buffers
        Xs = synthetic_leaves(3000, 5).astype(np.float32)
        seg = np.array([0, 100, 100, 1777, 3000], dtype=np.int64)
        for so in (None, seg):
            got = D.best_ucb_sharded(eng, Xs, Xs.shape[0], VS, so)
            exp = eng.best_ucb(Xs, VS, so)
            assert all(np.array_equal(a, b, equal_nan=True) for a, b in zip(got, exp))
        kids = tree.split_bounds([(0.0, 1.0)] * 5)
        boxes = np.array([kids[0], kids[2]])
        got = D.best_ucb_grow_sharded(eng, boxes, 6, VS)
        exp = eng.best_ucb_grow(boxes, 6, VS)
        assert all(np.array_equal(a, b) for a, b in zip(got, exp))
        with pytest.raises(ValueError):  # a rank must pass exactly its share of the batch
            D.best_ucb_sharded(eng, Xs[:10], Xs.shape[0], VS)
    finally:
        eng.comm_destroy()


def test_engine_group_of_one_device_matches_the_plain_engine():
    """``HipGPEngineGroup`` (one engine per device, each on its own thread, RCCL collectives inside the
    C-ABI calls) with a single device: every call goes through the group path."""
    from pygpso_amd.distributed import HipGPEngineGroup

    X, y = synthetic_problem(300, 5, seed=0)
    plain = _fitted("float64")
    grp = HipGPEngineGroup("float64", devices=[0])
    grp.set_data(X, y)
    f, g = grp.fit_eval("Matern52", [0.5], 1.0, 1e-3, float(y.mean()), want_grad=True)
    assert np.isfinite(f) and g.shape == (4,)
    Xs = synthetic_leaves(2500, 5)
    assert all(np.array_equal(a, b) for a, b in zip(grp.predict(Xs), plain.predict(Xs)))
    assert all(np.array_equal(a, b) for a, b in zip(grp.best_ucb(Xs, VS), plain.best_ucb(Xs, VS)))
    kids = tree.split_bounds([(0.0, 1.0)] * 5)
    boxes = np.array([kids[0], kids[2]])
    assert all(np.array_equal(a, b) for a, b in zip(grp.best_ucb_grow(boxes, 5, VS), plain.best_ucb_grow(boxes, 5, VS)))
    grp.close()


def test_G4_through_the_devices_argument():
    """``GPRSurrogate.default(devices=[...])``: GPSOptimiser unchanged, the surrogate's engine is the
    group.  (One device here -- the constructor path and the group engine under the optimiser loop.)"""
    from pygpso_amd import GPRSurrogate, GPSOptimiser, ParameterSpace
    from pygpso_amd.distributed import HipGPEngineGroup

    space = ParameterSpace(parameter_names=["x", "y"], parameter_bounds=G["G4"]["bounds"])
    surr = GPRSurrogate.default(devices=[0])
    surr.engine_factory = lambda: HipGPEngineGroup("float64", devices=[0])  # force the group engine on one GPU
    opt = GPSOptimiser(parameter_space=space, gp_surrogate=surr, exploration_depth=G["G4"]["depth"],
                       budget=G["G4"]["budget"])
    best = opt.run(rotated_peaks)
    assert isinstance(surr.gpflow_model.engine, HipGPEngineGroup)
    np.testing.assert_almost_equal(G["G4"]["best_coords"], best.normed_coord)
    assert np.around(best.score_mu, decimals=8) == G["G4"]["best_score"]


# ---- groups of 2, 3 and 8 ranks replayed on one device -------------------------------------------------------
def _replay_group(engines, world, Xs, seg):
    """rank r's half runs on engines[r % len(engines)] (every context holds the same posterior)."""
    from pygpso_amd.distributed import shard_range

    m = Xs.shape[0]
    payloads = [engines[r % len(engines)].shard_winners(r, world, Xs[slice(*shard_range(m, r, world))], m, VS, seg)
                for r in range(world)]
    nseg = 1 if seg is None else len(seg) - 1
    return engines[0].fold_winners(payloads, nseg, m, seg), payloads


@pytest.mark.parametrize("dtype,math", [("float64", None), ("mixed", "bf16x6"), ("mixed", "f16x3"), ("float32", "native")])
@pytest.mark.parametrize("world", [2, 3, 8])
def test_replayed_group_matches_the_single_context_call(dtype, math, world):
    """Leaf shards: multi-segment batches with empty segments and segments that straddle shard boundaries, a
    duplicate of the winner in a later shard and fewer leaves than ranks (empty shards) -- the folded
    result of `world` local halves is the single call's, bit for bit, indices included."""
    from pygpso_amd import HipGPEngine

    opts = {} if math is None else {"predict_math": math}
    root = _fitted(dtype, n=512, **opts)
    peer = HipGPEngine(dtype, **opts)
    _handoff(root, peer)  # the second context holds the posterior as a broadcast would leave it
    engines = [root, peer]
    m = 3001
    Xs = synthetic_leaves(m, 5, seed=3)
    i0 = int(root.best_ucb(Xs, VS)[0][0])
    Xs[(i0 + m // 2) % m] = Xs[i0]  # the winner again, in another shard: the lower index must win
    segs = [None, np.array([0, 100, 100, 1777, m], dtype=np.int64),
            np.array([0, 1, 2, m // world, m // world + 1, m - 1, m, m], dtype=np.int64)]
    for seg in segs:
        got, _ = _replay_group(engines, world, Xs, seg)
        exp = root.best_ucb(Xs, VS, seg)
        assert all(np.array_equal(a, b, equal_nan=True) for a, b in zip(got, exp)), (world, seg)
    few = Xs[: world - 1]  # the last rank's shard is empty
    got, payloads = _replay_group(engines, world, few, None)
    assert all(np.array_equal(a, b) for a, b in zip(got, root.best_ucb(few, VS)))
    assert np.isnan(payloads[-1][2]) and payloads[-1].view(np.int64)[3] == -1 and payloads[-1][-1] == 0.0


@pytest.mark.parametrize("world", [2, 3, 8])
@pytest.mark.parametrize("depth", [1, 2, 6, 8])
def test_replayed_group_growth_matches_the_single_context_call(world, depth):
    """On-device growth: every rank grows and scores its share of the reference rows of every box (closed-form
    slot ranges, appended near-duplicates of arbitrary boxes included); keys are global reference rows."""
    from pygpso_amd import HipGPEngine

    root = _fitted("float64")
    peer = HipGPEngine("float64")
    _handoff(root, peer)
    engines = [root, peer]
    rng = np.random.default_rng(depth)
    kids = tree.split_bounds([(0.0, 1.0)] * 5)
    lo = rng.random((2, 5)) * 0.5
    arbitrary = np.stack([lo, lo + 0.1 + 0.4 * rng.random((2, 5))], axis=2)  # boxes not cut from the unit cube
    boxes = np.concatenate([np.array([kids[0], kids[2]]), arbitrary])
    payloads = [engines[r % 2].shard_winners_grow(r, world, boxes, depth, VS) for r in range(world)]
    got = root.fold_winners(payloads, len(boxes))
    exp = root.best_ucb_grow(boxes, depth, VS)
    assert all(np.array_equal(a, b) for a, b in zip(got, exp))


@pytest.mark.parametrize("world", [2, 3, 8])
def test_fold_kernel_against_the_host_rule(world):
    """reduce_winners_kernel on crafted payloads -- NaN scores (np.argmax: the first NaN wins), ties across ranks
    (the lower global index wins), ranks without a winner, segment bases -- against the host statement of the rule."""
    from pygpso_amd import HipGPEngine
    from pygpso_amd.distributed import shard_range
    from tests.host_group import reduce_winners

    eng = HipGPEngine("float64")
    rng = np.random.default_rng(world)
    nseg, m = 7, 10007
    seg = np.sort(np.concatenate([[0, m], rng.integers(0, m + 1, nseg - 1)])).astype(np.int64)
    n_pay = 4 * nseg + 2
    for trial in range(40):
        pay = np.zeros((world, n_pay))
        rows = np.zeros((world, nseg, 4))  # host form: (ucb, global idx, mean, var)
        for r in range(world):
            lo, hi = shard_range(m, r, world)
            for s_ in range(nseg):
                plo, phi = np.clip(seg[s_], lo, hi), np.clip(seg[s_ + 1], lo, hi)  # this rank's piece of the segment
                ucb = rng.choice([np.nan, 1.0, 2.0, rng.random()], p=[0.15, 0.25, 0.25, 0.35])
                if phi <= plo or rng.random() < 0.1:
                    local, ucb = -1, np.nan  # no winner on this rank
                else:
                    local = int(rng.integers(0, phi - plo))
                mean, var = rng.random(), rng.random()
                pay[r, 4 * s_: 4 * s_ + 3] = (mean, var, ucb)
                pay[r].view(np.int64)[4 * s_ + 3] = local
                rows[r, s_] = (ucb, -1 if local < 0 else local + plo - seg[s_], mean, var)
        idx, mean, var, ucb = eng.fold_winners(list(pay), nseg, m, seg)
        for s_ in range(nseg):
            w = reduce_winners(rows[:, s_, :])
            assert int(idx[s_]) == int(w[1]), (trial, s_)
            if int(w[1]) >= 0:
                assert np.array_equal([ucb[s_], mean[s_], var[s_]], [w[0], w[2], w[3]], equal_nan=True)
    # growth payloads carry global reference rows: no bases
    pay = np.zeros((world, 6))
    for r in range(world):
        pay[r, :3] = (0.1 * r, 0.2, 5.0)  # the same score everywhere ...
        pay[r].view(np.int64)[3] = 1000 - r  # ... the LAST rank holds the lowest row
    idx, mean, _, _ = eng.fold_winners(list(pay), 1)
    assert int(idx[0]) == 1000 - (world - 1) and mean[0] == 0.1 * (world - 1)


@pytest.mark.parametrize("n", [300, 200])
def test_a_failed_half_reaches_every_rank(n):
    """Collective safety: a rank whose half fails still produces a payload (no winners, its status in the last
    slot) and the fold hands that status to every rank -- here a shard of the wrong size on one rank.  N = 200 pads to
    256: the shape at which plain best-UCB calls take the ONE-launch kernel, whose fallback flag shares the payload's
    status slot -- a group half must never be enqueued with it, and nothing of it may outlive a call whose result was
    only copied out (round-4 advice: finish_best read the group's verdict as that flag)."""
    from pygpso_amd import _lib as L
    from pygpso_amd.distributed import shard_range

    eng = _fitted("float64", n=n)
    Xs = synthetic_leaves(1000, 5)
    good = eng.shard_winners(0, 2, Xs[slice(*shard_range(1000, 0, 2))], 1000, VS)
    n_pay = good.shape[0]
    bad = np.empty(n_pay)
    rc = eng._lib.gpso_shard_winners(eng._h, 1, 2, Xs.ctypes.data, L.F64, L.MEM_HOST, 17, 1000, None, 1, VS,
                                     bad.ctypes.data_as(L._c_double_p))
    assert rc == L.E_ARG and bad[-1] == L.E_ARG and bad.view(np.int64)[3] == -1
    with pytest.raises(ValueError, match="payload of rank 1"):
        eng.fold_winners([good, bad], 1, 1000)
    # and the healthy group still folds afterwards
    ok = eng.shard_winners(1, 2, Xs[slice(*shard_range(1000, 1, 2))], 1000, VS)
    assert all(np.array_equal(a, b) for a, b in zip(eng.fold_winners([good, ok], 1, 1000), eng.best_ucb(Xs, VS)))
    # a growth call right behind payload-only calls is the plain call's (no stale one-launch state)
    box = np.array([[[0.1, 0.4]] * 5])
    eng.shard_winners(0, 2, Xs[slice(*shard_range(1000, 0, 2))], 1000, VS)
    got = eng.best_ucb_grow(box, 5, VS)
    ref = _fitted("float64", n=n).best_ucb_grow(box, 5, VS)
    assert all(np.array_equal(a, b) for a, b in zip(got, ref))


def test_precision_failure_is_a_group_verdict():
    """A float32 posterior at the reference's noise floor fails the self-test on the fitting rank: the broadcast
    returns GPSO_E_PRECISION (on every rank: the verdict is agreed before any buffer moves) instead of leaving the
    peers inside a collective, and ``GPRSurrogate(devices=[...], dtype="float32")`` escalates on the device."""
    from pygpso_amd import HipGPEngine
    from pygpso_amd import _lib as L
    from pygpso_amd import distributed as D
    from pygpso_amd.distributed import HipGPEngineGroup
    from pygpso_amd.model import HipGPR
    from pygpso_amd.kernels import Constant, Matern52

    X, y = synthetic_problem(200, 3, seed=0)
    theta = ("Matern52", [0.12], 3.0, 1.05e-6, float(y.mean()))
    eng = HipGPEngine("float32", tol_var=1e-7, tol_mean=1e-7)  # (tolerances no float engine meets: forces the verdict)
    eng.set_data(X, y)
    eng.fit_eval(*theta, want_grad=False)
    eng.comm_init(0, 1, D.exchange_unique_id(0, 1))
    try:
        with pytest.raises(L.GpsoPrecisionError):
            eng.broadcast_posterior(0)
        with pytest.raises(L.GpsoPrecisionError):
            eng.best_ucb_grow_sharded(np.array([[(0.0, 1.0)] * 3]), 3, VS)
        with pytest.raises(L.GpsoPrecisionError):
            eng.best_ucb_sharded(X, X.shape[0], VS)
    finally:
        eng.comm_destroy()
    # the group engine: options reach every rank; the verdict raises out of the group's threads ...
    grp = HipGPEngineGroup("float32", devices=[0])
    grp.set_tolerances(1e-7, 1e-7)
    grp.set_data(X, y)
    grp.fit_eval(*theta, want_grad=False)
    with pytest.raises(L.GpsoPrecisionError):
        grp.best_ucb(X, VS)
    grp.close()
    # ... and a model whose engine is a group escalates on the device (mixed, then float64)
    opts = {"tol_var": 1e-7, "tol_mean": 1e-7}
    model = HipGPR((X, y[:, None]), Matern52(lengthscales=0.12, variance=3.0), Constant(float(y.mean())),
                   noise_variance=1.05e-6, dtype="float64", engine_options=opts)
    model._open_engine = lambda dtype: HipGPEngineGroup(dtype, devices=[0], **opts)
    model.engine.close()
    model.engine = model._open_engine("float32")
    model.data = (X, y[:, None])
    mean, var = model.predict_y(X[:10])
    assert isinstance(model.engine, HipGPEngineGroup) and model.engine.dtype_name == "float64"
    ref = gpr.posterior(gpr.Theta(*theta), X, y)
    m_ref, v_ref = gpr.predict_y(ref, X[:10])
    np.testing.assert_allclose(np.asarray(mean)[:, 0], m_ref, rtol=1e-7, atol=1e-8)
    np.testing.assert_allclose(np.asarray(var)[:, 0], v_ref, rtol=1e-6, atol=1e-9)
    model.engine.close()


# ---- BASELINE configs C4 / C5 in their 8-GPU form, replayed on one device ---------------------------------------
# All of the group call except the ncclAllGather itself runs here AT SIZE: 8 local halves (gpso_shard_winners on two
# contexts that hold the posterior -- the second one received it as a broadcast would deliver it: one contiguous range) +
# gpso_fold_winners, against ONE gpso_best_ucb over the whole batch (bit-identical, indices included) and the float64
# oracle on a sub-sample.
def _replay_at_size(root, Xs, segs, world=8):
    from pygpso_amd import HipGPEngine

    peer = HipGPEngine(root.dtype_name)
    nb = _handoff_span(root, peer)
    assert peer.posterior_hash() == root.posterior_hash()
    whole = {}
    for key, seg in segs.items():
        whole[key] = root.best_ucb(Xs, VS, seg)
        got, payloads = _replay_group([root, peer], world, Xs, seg)
        assert all(np.array_equal(a, b, equal_nan=True) for a, b in zip(got, whole[key])), key
        assert all(p[-1] == 0.0 for p in payloads)
    peer.close()
    return whole, nb


def test_config_C4_full_batch_replayed_world8():
    """C4 as BASELINE.json states it: D = 20, N_train = 8192, ALL 262 144 leaves, fp32, 8 ranks."""
    from pygpso_amd import HipGPEngine
    from tests.test_gpu_parity import FLOAT_BOUNDS, _c4_posterior, _fit, _problem

    n, d, m = 8192, 20, 262144
    X, y, th = _problem(n, d, variance=1.0)
    post = _c4_posterior(th, X, y)
    root = HipGPEngine("float32")
    _fit(root, X, y, th, grad=False)
    assert root.precision_info()["predict_math"] == "f16x3"
    Xs = synthetic_leaves(m, d).astype(np.float32)
    segs = {"one": None, "ragged": np.array([0, 1000, 1000, m // 8 + 5, m // 2, m - 1, m], dtype=np.int64)}
    whole, nb = _replay_at_size(root, Xs, segs)
    assert n * n * 4 < nb < 1.02 * n * n * 4  # two fp16 planes (2 x 2 bytes per entry) + the small buffers; the packed f32 factor stays home
    # float64 oracle: a 128-leaf sub-sample and the winner itself
    sub = np.random.default_rng(9).choice(m, 127, replace=False)
    sub = np.append(sub, int(whole["one"][0][0]))
    mean, var = root.predict(Xs[sub])
    mean_ref, var_ref = gpr.predict_y(post, Xs[sub].astype(np.float64))
    em, ev = np.max(np.abs(mean - mean_ref)) / np.max(np.abs(y)), np.max(np.abs(var - var_ref)) / th.variance
    print(f"C4 full batch: |d mean| {em:.2e} max|y|, |d var| {ev:.2e} sigma^2")
    assert em <= FLOAT_BOUNDS["C4"][0] and ev <= FLOAT_BOUNDS["C4"][1]
    assert whole["one"][1][0] == mean[-1] and whole["one"][2][0] == var[-1]  # the call's winner values are predict's
    # growth at depth 12 (265 720 reference rows per box, SURVEY 8a6's C4 row) through the sharded halves
    kids = tree.split_bounds([(0.0, 1.0)] * d)
    boxes = np.array([kids[0], kids[2]])
    exp = root.best_ucb_grow(boxes, 12, VS)
    assert root.last_count(1) == 2 * 265720 and root.last_count(0) == 2 * 3 ** 11
    for world in (8, 3):
        payloads = [root.shard_winners_grow(r, world, boxes, 12, VS) for r in range(world)]
        got = root.fold_winners(payloads, 2)
        assert all(np.array_equal(a, b) for a, b in zip(got, exp)), world


def test_config_C5_full_batch_replayed_world8():
    """C5 as BASELINE.json states it: D = 40, N_train = 16384, ALL 1 048 576 leaves (the config's "bf16 Gram" is run
    as fp32 Gram + fp32 Cholesky + 16-bit-split apply), 8 ranks."""
    from pygpso_amd import HipGPEngine
    from tests.test_gpu_parity import FLOAT_BOUNDS, _c5_reference, _fit

    n, d, m = 16384, 40, 1048576
    ref = _c5_reference(1e-3)
    X, y, th = ref["X"], ref["y"], ref["th"]
    root = HipGPEngine("float32")
    f, _ = _fit(root, X, y, th, grad=False)
    assert abs(f - ref["nlml"]) <= 1e-4 * abs(ref["nlml"])
    assert root.precision_info()["predict_math"] == "f16x3"
    Xs = synthetic_leaves(m, d).astype(np.float32)
    segs = {"one": None, "ragged": np.array([0, 7, m // 8 - 1, m // 8 + 1, m // 2, m], dtype=np.int64)}
    whole, nb = _replay_at_size(root, Xs, segs)
    assert 1.0e9 < nb < 1.1e9  # 2 fp16 planes of 16384^2 = 1.07 GB is what a broadcast moves
    sub = ref["sub"]  # (the first 131 072 leaves of the larger batch are the share test's leaves)
    mean, var = root.predict(Xs[sub])
    em = np.max(np.abs(mean - ref["mean_ref"])) / max(1.0, np.max(np.abs(y)))
    ev = np.max(np.abs(var - ref["var_ref"])) / th.variance
    print(f"C5 full batch: |d mean| {em:.2e} max|y|, |d var| {ev:.2e} sigma^2")
    assert em <= FLOAT_BOUNDS["C5"][0] and ev <= FLOAT_BOUNDS["C5"][1]
    i0 = int(whole["one"][0][0])
    m1, v1 = root.predict(Xs[i0:i0 + 1])
    assert whole["one"][1][0] == m1[0] and whole["one"][2][0] == v1[0]


# ---- round 6: an appended posterior is handed on by ROWS (gpso_posterior_dirty_ranges / gpso_broadcast_posterior_rows) ---------
def _copy_ranges(src, dst, ranges):
    import torch

    dev = torch.device("cuda", src.device)
    ptr, off, _nb = src.posterior_span()
    moved = 0
    for o, nb in ranges:
        a = torch.as_tensor(_DeviceBytes(ptr + (o - off), nb), device=dev)
        b = torch.as_tensor(_DeviceBytes(dst.posterior_span_at(o - o % 256, nb + o % 256) + o % 256, nb), device=dev)
        b.copy_(a)
        moved += nb
    torch.cuda.synchronize()
    return moved


@pytest.mark.parametrize("dtype,math", [("float64", None), ("mixed", "f16x3"), ("float32", "auto"), ("mixed", "bf16x6"),
                                        ("float32", "native")])
def test_an_append_is_handed_on_by_rows(dtype, math):
    """What gpso_broadcast_posterior_rows moves, spelled out with device-to-device copies on two contexts of one GPU: after a
    full hand-off, an append on the sender dirties <= 9 small ranges of the arena (hyper block, the new rows of the scaled
    inputs / their fragments / norms, alpha, the tile rows of each predict-ready copy of L^-1 that hold new rows); copying
    THOSE gives the receiver the fingerprint and the prediction bits of a full hand-off of the extended posterior."""
    from pygpso_amd import HipGPEngine

    n, d, ks = 1000, 5, (5, 1, 17)
    X, y = synthetic_problem(n + sum(ks), d, seed=0)
    opts = {} if math is None else {"predict_math": math}
    a, b, c = (HipGPEngine(dtype, **opts) for _ in range(3))
    a.set_data(X[:n], y[:n])
    a.fit_eval("Matern52", [0.5], 1.0, 1e-3, float(y.mean()), want_grad=False)
    assert len(a.posterior_dirty_ranges()) == 1  # nobody holds anything yet: the whole span
    whole = _handoff_span(a, b)
    a.posterior_mark_synced()
    assert a.posterior_dirty_ranges() == []
    Xs = synthetic_leaves(2500, d)
    lo = n
    for k in ks:
        _, in_place = a.append(X[lo:lo + k], y[lo:lo + k])
        assert in_place
        lo += k
        if k == 1:
            continue  # (two appends travel together)
        ranges = a.posterior_dirty_ranges()
        assert 5 <= len(ranges) <= 9, ranges
        moved = _copy_ranges(a, b, ranges)
        b.adopt_posterior()
        a.posterior_mark_synced()
        print(f"{dtype}/{math}: {len(ranges)} ranges, {moved} bytes instead of {whole} ({whole / moved:.0f}x less)")
        assert moved < 0.12 * whole and b.n == a.n == lo
        _handoff_span(a, c)  # the whole extended posterior, for comparison
        assert b.posterior_hash() == c.posterior_hash() == a.posterior_hash()
        for e in (b, c):
            assert all(np.array_equal(u, v) for u, v in zip(a.best_ucb(Xs, VS), e.best_ucb(Xs, VS)))
        pa, pb = a.predict(Xs), b.predict(Xs)
        assert np.array_equal(pa[0], pb[0]) and np.array_equal(pa[1], pb[1])
        assert a.posterior_dirty_ranges() == []
    # a new fit: another posterior -- the whole span again
    a.fit_eval("Matern52", [0.6], 1.1, 1e-3, float(y.mean()), want_grad=False)
    assert len(a.posterior_dirty_ranges()) == 1


def test_a_crossed_fp16_scale_or_another_predict_math_sends_the_whole_span():
    """The fp16 planes are scaled by a power of two that follows max |L^-1|.  Sparse points (D = 12, short lengthscale: every
    conditional variance ~ s2, max |L^-1| ~ 1) and then a new point on top of an old one (Schur complement 2 x noise:
    1 / L22 ~ 70): the scale crosses six powers of two, every row's planes are repacked -- the hand-off is the whole span."""
    from pygpso_amd import HipGPEngine

    n, d = 300, 12
    X, y = synthetic_problem(n + 2, d, seed=17)
    X[n] = X[3] + 1e-7
    y[n] = y[3]
    a, b = HipGPEngine("mixed", predict_math="f16x3", precision_check=False), HipGPEngine("mixed", predict_math="f16x3", precision_check=False)
    a.set_data(X[:n], y[:n])
    a.fit_eval("Matern52", [0.2], 1.0, 1e-4, float(y.mean()), want_grad=False)
    whole = _handoff_span(a, b)
    a.posterior_mark_synced()
    _, in_place = a.append(X[n:n + 1], y[n:n + 1])
    assert in_place
    ranges = a.posterior_dirty_ranges()
    assert len(ranges) == 1 and ranges[0][1] == whole, ranges
    _copy_ranges(a, b, ranges)
    b.adopt_posterior()
    a.posterior_mark_synced()
    Xs = synthetic_leaves(1500, d)
    assert all(np.array_equal(u, v) for u, v in zip(a.best_ucb(Xs, VS), b.best_ucb(Xs, VS)))
    post = gpr.posterior(gpr.Theta("Matern52", np.array([0.2]), 1.0, 1e-4, float(y.mean())), X[:n + 1], y[:n + 1])
    mean, var = b.predict(Xs)
    mean_ref, var_ref = gpr.predict_y(post, Xs)
    assert np.max(np.abs(mean - mean_ref)) <= 6e-4 * np.max(np.abs(y)) and np.max(np.abs(var - var_ref)) <= 1e-4
    a.append(X[n + 1:], y[n + 1:])  # an ordinary point: rows again
    assert len(a.posterior_dirty_ranges()) > 1
    a.set_predict_math("bf16x6")  # other pieces than the peer holds
    assert len(a.posterior_dirty_ranges()) == 1


def test_rows_hand_off_on_a_real_communicator_of_one_rank():
    """gpso_broadcast_posterior_rows on an RCCL communicator (world 1: the header, the verdict all-reduce and the grouped
    ncclBroadcasts of the ranges all execute): rows after an append, the whole after a fit or without a previous hand-off."""
    from pygpso_amd import distributed as D

    n, d, k = 700, 5, 6
    X, y = synthetic_problem(n + k, d, seed=3)
    for dtype in ("float32", "float64"):
        from pygpso_amd import HipGPEngine

        eng = HipGPEngine(dtype)
        eng.set_data(X[:n], y[:n])
        eng.fit_eval("Matern52", [0.5], 1.0, 1e-3, float(y.mean()), want_grad=False)
        eng.comm_init(0, 1, D.exchange_unique_id(0, 1))
        try:
            assert eng.broadcast_posterior_rows(0) is False  # no previous hand-off: the whole range
            whole = eng.last_count(0)
            Xs = synthetic_leaves(1000, d)
            before = eng.best_ucb(Xs, VS)
            eng.append(X[n:], y[n:])
            assert eng.broadcast_posterior_rows(0) is True
            rows = eng.last_count(0)
            print(f"{dtype}: rows hand-off {rows} bytes, whole {whole}")
            assert 0 < rows < 0.15 * whole and eng.n == n + k
            assert eng.broadcast_posterior_rows(0) is True and eng.last_count(0) == 0  # up to date: nothing moves
            got = D.best_ucb_sharded(eng, Xs, Xs.shape[0], VS)
            ref = HipGPEngine(dtype)
            ref.set_data(X[:n], y[:n])
            ref.fit_eval("Matern52", [0.5], 1.0, 1e-3, float(y.mean()), want_grad=False)
            ref.append(X[n:], y[n:])
            assert all(np.array_equal(u, v) for u, v in zip(got, ref.best_ucb(Xs, VS)))
            assert not np.array_equal(before[3], got[3])
            eng.fit_eval("Matern52", [0.6], 1.0, 1e-3, float(y.mean()), want_grad=False)
            assert eng.broadcast_posterior_rows(0) is False
        finally:
            eng.comm_destroy()

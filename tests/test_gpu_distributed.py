"""GPU checks of the multi-GPU plumbing that can run on ONE device: the posterior hand-off protocol on
real device memory (what a broadcast does, spelled out with device-to-device copies), and the RCCL
group calls of the C-ABI in a world of size 1 -- per-rank style and the thread-driven
``HipGPEngineGroup`` behind ``GPRSurrogate(devices=[...])``.  (World sizes > 1: the world-2 ``gloo``
tests of tests/test_distributed_cpu.py cover the sharding logic; the driver's scaling run covers RCCL.)"""
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
    assert _handoff(src, dst) == 6
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

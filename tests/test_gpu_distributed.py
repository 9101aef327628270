"""GPU checks of the multi-GPU plumbing that can run on ONE device: the posterior hand-off
protocol on real device memory, and the RCCL ("nccl") calls in a world of size 1."""
import socket

import numpy as np
import pytest

from oracle import gpr
from tests.helpers import synthetic_leaves, synthetic_problem

pytestmark = pytest.mark.gpu
VS = gpr.VARSIGMA_DEFAULT


def _fitted(dtype):
    from pygpso_amd import HipGPEngine

    X, y = synthetic_problem(300, 5, seed=0)
    eng = HipGPEngine(dtype)
    eng.set_data(X, y)
    eng.fit_eval("Matern52", [0.5], 1.0, 1e-3, float(y.mean()), want_grad=False)
    return eng


def test_posterior_handoff_carries_the_bf16_pieces():
    import torch

    from pygpso_amd import HipGPEngine
    from pygpso_amd import distributed as D

    X, y = synthetic_problem(512, 5, seed=0)
    src = HipGPEngine("float32", predict_math="bf16x6")
    src.set_data(X, y)
    src.fit_eval("Matern52", [0.5], 1.0, 1e-3, float(y.mean()), want_grad=False)
    dst = HipGPEngine("float32", predict_math="bf16x6")
    dst.alloc_posterior(src.n, src.d)
    a, b = D.engine_posterior_tensors(src), D.engine_posterior_tensors(dst)
    assert len(a) == len(b) == 6
    for s_, t_ in zip(a, b):
        t_.copy_(s_)
    torch.cuda.synchronize()
    dst.adopt_posterior()
    Xs = synthetic_leaves(1500, 5)
    assert all(np.array_equal(p, q) for p, q in zip(src.predict(Xs), dst.predict(Xs)))


@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_posterior_handoff_between_two_contexts(dtype):
    """What a broadcast does, spelled out with device-to-device copies: the receiver predicts
    bit-identically to the rank that fitted."""
    from pygpso_amd import HipGPEngine
    from pygpso_amd import distributed as D

    src = _fitted(dtype)
    dst = HipGPEngine(dtype)
    dst.alloc_posterior(src.n, src.d)
    a, b = D.engine_posterior_tensors(src), D.engine_posterior_tensors(dst)
    assert [t.numel() for t in a] == [t.numel() for t in b] and len(a) == 5
    for s, t in zip(a, b):
        t.copy_(s)
    import torch

    torch.cuda.synchronize()
    dst.adopt_posterior()
    Xs = synthetic_leaves(2000, 5)
    m1, v1 = src.predict(Xs)
    m2, v2 = dst.predict(Xs)
    assert np.array_equal(m1, m2) and np.array_equal(v1, v2)
    assert all(np.array_equal(p, q) for p, q in zip(src.best_ucb(Xs, VS), dst.best_ucb(Xs, VS)))


def test_nccl_world_of_one():
    import torch
    import torch.distributed as dist

    from pygpso_amd import distributed as D

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        eng = _fitted("float32")
        assert D.can_view_engine_memory(eng) is True
        D.broadcast_posterior(eng, src=0)  # RCCL broadcast straight on the library's buffers
        Xs = synthetic_leaves(3000, 5)
        leaves = torch.from_numpy(Xs.astype(np.float32)).cuda()
        got = D.best_ucb_sharded(eng, leaves, 0, VS)
        idx, mean, var, ucb = eng.best_ucb(leaves, VS)
        assert got == (int(idx[0]), float(mean[0]), float(var[0]), float(ucb[0]))
    finally:
        dist.destroy_process_group()

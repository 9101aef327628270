"""
World-size-2 ``gloo`` tests of the multi-GPU path (leaf sharding, posterior broadcast protocol,
winner reduction) on CPU, with the oracle-backed test double as the per-rank engine.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, m, dup, out):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist

    from oracle import gpr
    from pygpso_amd import distributed as D
    from tests.helpers import synthetic_leaves, synthetic_problem
    from tests.oracle_engine import OracleEngine

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        X, y = synthetic_problem(80, 3, seed=0)
        leaves = synthetic_leaves(m, 3, seed=1)
        if dup:  # the global winner also appears, later, in the other rank's shard
            eng0 = OracleEngine()
            eng0.set_data(X, y)
            eng0.fit_eval("Matern52", [0.4], 1.0, 1e-3, 0.0, want_grad=False)
            i0 = int(eng0.best_ucb(leaves, gpr.VARSIGMA_DEFAULT)[0][0])
            leaves[(i0 + m // 2) % m] = leaves[i0]
        eng = OracleEngine()
        if rank == 0:  # only the root fits
            eng.set_data(X, y)
            eng.fit_eval("Matern52", [0.4], 1.0, 1e-3, 0.0, want_grad=False)
        assert D.can_view_engine_memory(eng) is True  # agreed across ranks before any bulk collective
        D.broadcast_posterior(eng, src=0)
        lo, hi = D.shard_range(m, rank, world)
        got = D.best_ucb_sharded(eng, leaves[lo:hi], lo, gpr.VARSIGMA_DEFAULT)
        # single-process answer on the whole batch
        ref = OracleEngine()
        ref.set_data(X, y)
        ref.fit_eval("Matern52", [0.4], 1.0, 1e-3, 0.0, want_grad=False)
        i, mu, vv, uu = ref.best_ucb(leaves, gpr.VARSIGMA_DEFAULT)
        out[rank] = (got, (int(i[0]), float(mu[0]), float(vv[0]), float(uu[0])))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("m,dup", [(1001, False), (640, True), (3, False)])
def test_two_ranks_agree_with_single_process(m, dup):
    world = 2
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_worker, args=(world, _free_port(), m, dup, out), nprocs=world, join=True)
        res = dict(out)
    assert sorted(res) == [0, 1]
    assert res[0][0] == res[1][0]  # every rank holds the same winner
    for got, ref in res.values():
        assert got == ref  # bit-identical to the unsharded result, first-max tie rule included


def test_shard_ranges_partition_the_batch():
    from pygpso_amd.distributed import shard_range

    for m in (0, 1, 7, 8, 65536, 1000003):
        for world in (1, 2, 3, 8):
            spans = [shard_range(m, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == m
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def test_reduce_winners_first_max_and_nan_rules():
    from pygpso_amd.distributed import reduce_winners

    rows = np.array([[1.0, 10, 0, 0], [2.0, 700, 0, 0], [2.0, 300, 0, 0], [0.5, 2, 0, 0]])
    assert int(reduce_winners(rows)[1]) == 300  # tie -> lowest global index
    rows = np.array([[1.0, 10, 0, 0], [np.nan, 50, 0, 0], [np.nan, 40, 0, 0]])
    assert int(reduce_winners(rows)[1]) == 40  # NaN counts as the maximum, first one wins
    rows = np.array([[np.nan, -1, 0, 0], [0.1, 5, 0, 0]])
    assert int(reduce_winners(rows)[1]) == 5  # an empty shard never wins

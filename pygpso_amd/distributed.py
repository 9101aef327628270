"""
Multi-GPU leaf-UCB: one process per GPU, ``torch.distributed`` over RCCL/xGMI ("nccl" backend).

The predict path shards naturally: leaves are independent given the
posterior.  So
  * the GP is fitted on ONE rank and its predict-ready state (tile-packed L^-1, scaled inputs,
    norms, alpha, hyper-parameter block -- ``gpso_posterior_buffers``) is BROADCAST to the peers:
    the only bulk collective, once per fit; a 1 -> 7 broadcast drives all 7 xGMI links of the root;
  * every rank scores its own contiguous range of the leaf batch with no data-path collective;
  * the per-rank winners (4 doubles each) are all-gathered and reduced with the same first-max
    rule ``np.argmax`` applies to the unsharded batch, so the sharded result is bit-identical to
    the single-GPU one.
Nothing is all-reduced.  The reference has no distributed path at all.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist


class _DeviceBytes:
    """Expose a raw device allocation to torch through ``__cuda_array_interface__``."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {
            "shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False), "version": 2,
            "strides": None,
        }


def device_bytes_as_tensor(ptr, nbytes, device):
    """uint8 view (no copy) of ``nbytes`` of device memory at ``ptr`` on ``cuda:device``."""
    return torch.as_tensor(_DeviceBytes(ptr, nbytes), device=torch.device("cuda", device))


def engine_posterior_tensors(engine):
    """The predict-ready state of ``engine`` as a list of flat tensors that can be broadcast."""
    if hasattr(engine, "posterior_tensors"):  # engines that keep their state in torch tensors already
        return engine.posterior_tensors()
    return [device_bytes_as_tensor(p, nb, engine.device) for p, nb in engine.posterior_buffers()]


def shard_range(m, rank, world):
    """Contiguous leaf range [lo, hi) of ``rank``: global row order is preserved across ranks."""
    base, extra = divmod(int(m), int(world))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def can_view_engine_memory(engine, group=None):
    """True on every rank iff EVERY rank can hand the engine's device buffers to torch (agreed with
    a MIN all-reduce on a torch-allocated tensor, so a rank that cannot does not leave the others
    waiting inside a broadcast).  Callers that get False re-fit on each rank instead."""
    ok = 1
    try:
        if hasattr(engine, "posterior_tensors"):
            pass  # the engine already keeps its state in torch tensors
        elif engine.n > 0:
            for t in engine_posterior_tensors(engine):
                _ = t.numel()
        else:
            probe = device_bytes_as_tensor(torch.zeros(16, dtype=torch.uint8, device=torch.device(
                "cuda", engine.device)).data_ptr(), 16, engine.device)
            _ = probe.numel()
    except Exception:  # noqa: BLE001 - any failure means "do not broadcast"
        ok = 0
    flag = torch.tensor([ok], dtype=torch.int32)
    if dist.get_backend(group) == "nccl":
        flag = flag.cuda(engine.device)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
    return bool(int(flag.item()))


def broadcast_posterior(engine, src=0, group=None):
    """Make the posterior resident on rank ``src`` resident on every rank of ``group``."""
    rank = dist.get_rank(group)
    shape = torch.tensor([engine.n, engine.d] if rank == src else [0, 0], dtype=torch.int64)
    backend = dist.get_backend(group)
    if backend == "nccl":
        shape = shape.cuda(engine.device)
    dist.broadcast(shape, src=src, group=group)
    n, d = (int(v) for v in shape.cpu())
    if rank != src:
        engine.alloc_posterior(n, d)
    for t in engine_posterior_tensors(engine):
        dist.broadcast(t, src=src, group=group)
    if backend == "nccl":
        torch.cuda.synchronize(engine.device)
    if rank != src:
        engine.adopt_posterior()


def _better(a, b):
    """np.argmax order on (ucb, global index): NaN is the maximum, ties go to the lower index."""
    (ua, ia), (ub, ib) = a, b
    if ia < 0:
        return False
    if ib < 0:
        return True
    na, nb = np.isnan(ua), np.isnan(ub)
    if na != nb:
        return bool(na)
    if not na and ua != ub:
        return bool(ua > ub)
    return ia < ib


def reduce_winners(rows):
    """rows [world, 4] of (ucb, global idx, mean, var) -> the winning row."""
    best = None
    for r in rows:
        if best is None or _better((r[0], int(r[1])), (best[0], int(best[1]))):
            best = r
    return best


class _WinnerExchange:
    """Preallocated buffers for the per-predict all-gather of (ucb, global idx, mean, var): pinned
    host staging + device tensors, one collective and ONE stream synchronisation per step."""

    def __init__(self, engine, group):
        self.world = dist.get_world_size(group)
        self.nccl = dist.get_backend(group) == "nccl"
        if self.nccl:
            dev = torch.device("cuda", engine.device)
            self.mine_host = torch.empty(4, dtype=torch.float64).pin_memory()
            self.rows_host = torch.empty(self.world * 4, dtype=torch.float64).pin_memory()
            self.mine_dev = torch.empty(4, dtype=torch.float64, device=dev)
            self.rows_dev = torch.empty(self.world * 4, dtype=torch.float64, device=dev)
        else:
            self.mine_host = torch.empty(4, dtype=torch.float64)
            self.rows_host = torch.empty(self.world * 4, dtype=torch.float64)

    def exchange(self, ucb, gidx, mean, var, group):
        m = self.mine_host
        m[0], m[1], m[2], m[3] = ucb, gidx, mean, var
        if self.nccl:
            self.mine_dev.copy_(m, non_blocking=True)
            dist.all_gather_into_tensor(self.rows_dev, self.mine_dev, group=group)
            self.rows_host.copy_(self.rows_dev, non_blocking=True)
            torch.cuda.current_stream(self.mine_dev.device).synchronize()
        else:
            dist.all_gather_into_tensor(self.rows_host, m, group=group)
        return self.rows_host.numpy().reshape(self.world, 4)


_exchanges = {}


def best_ucb_sharded(engine, local_leaves, offset, varsigma, group=None):
    """Score this rank's leaf shard (rows ``offset ..`` of the global batch) and agree on the global
    winner.  Returns (global_idx, mean, var, ucb) -- identical on every rank, and identical to
    ``engine.best_ucb(all_leaves)`` on one GPU."""
    idx, mean, var, ucb = engine.best_ucb(local_leaves, varsigma)
    gidx = float(idx[0] + offset) if idx[0] >= 0 else -1.0  # < 2^53: exact in float64
    key = (id(engine), id(group))
    ex = _exchanges.get(key)
    if ex is None:
        ex = _exchanges[key] = _WinnerExchange(engine, group)
    w = reduce_winners(ex.exchange(float(ucb[0]), gidx, float(mean[0]), float(var[0]), group))
    return int(w[1]), float(w[2]), float(w[3]), float(w[0])

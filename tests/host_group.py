"""Host mirror of the group protocol the C-ABI runs with RCCL (api.hip: best_ucb_sharded / best_ucb_grow_sharded,
predict.hip: reduce_winners_kernel), over a caller-supplied transport: TEST infrastructure for the world-size-2
``gloo`` tests on CPU, where the per-rank engine is the oracle-backed double (tests/oracle_engine.py)."""
import numpy as np

from pygpso_amd.distributed import shard_range


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


class HostGroup:
    """Sharding + winner rule on the host for an engine object that has no RCCL communicator of its own.
    ``allgather(a)``: float64 array -> [world, *a.shape] stacked in rank order; ``bcast(obj, src)``: python object
    from ``src`` to all."""

    def __init__(self, engine, rank, world, allgather, bcast):
        self.engine, self.rank, self.world = engine, int(rank), int(world)
        self._allgather, self._bcast = allgather, bcast

    def broadcast_posterior(self, src=0):
        state = self.engine.export_posterior() if self.rank == src else None
        state = self._bcast(state, src)
        if self.rank != src:
            self.engine.import_posterior(state)
        self.engine.sync_n = self.engine.n  # every rank: the group holds these rows (api.hip: posterior_mark_synced)
        self.last_bytes = sum(np.asarray(v).nbytes for v in state.values() if isinstance(v, np.ndarray))

    def broadcast_posterior_rows(self, src=0):
        """Host mirror of gpso_broadcast_posterior_rows: the root offers the rows since the last hand-off (or nothing
        but the whole), every rank says whether it holds that base (min over the ranks), then either the rows travel
        or -- on EVERY rank -- the whole posterior.  Returns True when rows sufficed."""
        eng = self.engine
        offer = None
        if self.rank == src:
            offer = (eng.sync_n if (eng.post is not None and 0 <= eng.sync_n <= eng.n) else -1, eng.n)
        n_base, n_now = self._bcast(offer, src)
        mine = 1.0 if n_base >= 0 and (self.rank == src or (eng.post is not None and eng.sync_n == n_base and eng.n == n_base)) else 0.0
        if float(np.min(self._allgather(np.array([mine])))) != 1.0:
            self.broadcast_posterior(src)
            return False
        state = eng.export_rows(n_base) if self.rank == src else None
        state = self._bcast(state, src)
        if self.rank != src and n_now > n_base:
            eng.import_rows(state)
        eng.sync_n = eng.n
        self.last_bytes = sum(np.asarray(v).nbytes for v in state.values() if isinstance(v, np.ndarray))
        return True

    def _fold(self, mine):
        """mine [nseg, 4] = (ucb, global idx or -1, mean, var) -> winners [nseg, 4]"""
        rows = np.asarray(self._allgather(np.ascontiguousarray(mine, dtype=np.float64)))
        return np.stack([reduce_winners(rows[:, s, :]) for s in range(mine.shape[0])])

    def best_ucb_sharded(self, local_leaves, m_global, varsigma, seg_off=None):
        lo, hi = shard_range(m_global, self.rank, self.world)
        assert local_leaves.shape[0] == hi - lo
        so = np.array([0, m_global], dtype=np.int64) if seg_off is None else np.asarray(seg_off, dtype=np.int64)
        local = np.clip(so, lo, hi) - lo
        idx, mean, var, ucb = self.engine.best_ucb(local_leaves, varsigma, local)
        # index relative to the GLOBAL segment start
        gidx = np.where(idx >= 0, idx + (np.clip(so[:-1], lo, hi) - so[:-1]), -1).astype(np.float64)
        w = self._fold(np.stack([ucb, gidx, mean, var], axis=1))
        return w[:, 1].astype(np.int64), w[:, 2], w[:, 3], w[:, 0]

    def best_ucb_grow_sharded(self, bounds, depth, varsigma):
        b = np.asarray(bounds, dtype=np.float64)
        if b.ndim == 2:
            b = b[None]
        rows = self.engine.grow_rows(depth)
        lo, hi = shard_range(rows, self.rank, self.world)
        grown = self.engine.grow(b, depth)[:, lo:hi, :]  # this rank's reference rows of every box
        nseg = b.shape[0]
        flat = grown.reshape(nseg * (hi - lo), b.shape[1])
        idx, mean, var, ucb = self.engine.best_ucb(flat, varsigma, np.arange(nseg + 1) * (hi - lo))
        gidx = np.where(idx >= 0, idx + lo, -1).astype(np.float64)
        w = self._fold(np.stack([ucb, gidx, mean, var], axis=1))
        return w[:, 1].astype(np.int64), w[:, 2], w[:, 3], w[:, 0]

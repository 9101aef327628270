"""
Multi-GPU leaf-UCB: one ``HipGPEngine`` per MI355X, joined into a group through RCCL **behind the
C-ABI** (``gpso_comm_init`` / ``gpso_broadcast_posterior`` / ``gpso_best_ucb_sharded`` /
``gpso_best_ucb_grow_sharded``, include/gpso_hip.h).  This module holds no torch and no collective of
its own: it bootstraps the group id, computes the shard ranges and calls the library.

The predict path shards naturally: leaves are independent given the posterior.
  * the GP is fitted on ONE rank and its predict-ready state (triangle-packed L^-1, scaled inputs,
    norms, alpha, hyper-parameter block) is BROADCAST to the peers over xGMI: the only bulk collective,
    once per fit;
  * every rank scores its own contiguous share of the leaf batch -- or, for on-device growth, generates
    and scores its share of the reference rows of every box -- with no data-path collective;
  * the per-segment winners (4 doubles each) are all-gathered and folded on the device with the same
    first-max rule ``np.argmax`` applies to the unsharded batch, so the sharded result is bit-identical
    to the single-GPU one and identical on every rank.
Nothing is all-reduced.  The reference has no distributed path at all.

Two ways to form a group:
  * one process per GPU (``torchrun`` / MPI style): every process creates its engine, rank 0 calls
    ``unique_id()`` and ships the 128 bytes to the others by the launcher's rendezvous
    (``exchange_unique_id``: a TCP exchange on MASTER_ADDR / MASTER_PORT, no framework needed), then all
    call ``engine.comm_init(rank, world, uid)``;
  * ONE process, several GPUs (``GPRSurrogate(devices=[...])``): ``HipGPEngineGroup`` drives one engine
    per device from its own thread (ctypes releases the GIL; RCCL supports one communicator per thread).

What a group of N ranks computes can be replayed on ONE device through the two halves of the sharded calls
(``HipGPEngine.shard_winners`` / ``shard_winners_grow`` / ``fold_winners``: the very kernels and index arithmetic
of the group calls, the all-gather replaced by a concatenation) -- tests/test_gpu_distributed.py does so for
world sizes 2, 3 and 8; the host mirror of the protocol used by the world-size-2 ``gloo`` tests on CPU lives
with the tests (tests/host_group.py).
"""
from __future__ import annotations

import ctypes as C
import os
import socket
import struct
import time
from concurrent.futures import FIRST_EXCEPTION, ThreadPoolExecutor, wait

import numpy as np

from . import _lib as L


# -- group bootstrap ------------------------------------------------------------------------------
def unique_id():
    """128-byte RCCL group id (``ncclGetUniqueId`` through the C-ABI); call on ONE rank."""
    lib = L.load()
    buf = C.create_string_buffer(L.UNIQUE_ID_BYTES)
    rc = lib.gpso_comm_unique_id(buf)
    if rc != L.OK:
        raise L.GpsoHipError(rc, lib.gpso_last_error(None).decode())
    return bytes(buf.raw)


def _id_token():
    """What a rank must present to be served the group id: the launcher's run id when there is one
    (TORCHELASTIC_RUN_ID), so that a stray connection to the port is neither served nor counted."""
    return (os.environ.get("GPSO_GROUP_TOKEN") or os.environ.get("TORCHELASTIC_RUN_ID") or "gpso").encode()[:64]


def exchange_unique_id(rank, world, addr=None, port=None, timeout=120.0, make_id=None, grace=0.25):
    """Rank 0 creates the id and serves it to the other ranks over TCP on (addr, port) -- by default
    MASTER_ADDR and MASTER_PORT + 1 of the launcher's environment (the port itself belongs to the
    launcher's own store).  A client introduces itself with its rank and the group token; rank 0 answers
    every rank 1..world-1 and ignores anything else (a port scanner, a wrong token); once every distinct rank
    has been answered the port stays open for ``grace`` more seconds, so that a rank whose answer was lost can
    ask again.  The token is the launcher's run id under torchrun (TORCHELASTIC_RUN_ID); outside torchrun set
    GPSO_GROUP_TOKEN to something private to the job -- the fallback "gpso" only keeps strangers out by accident."""
    make_id = make_id or unique_id
    if world == 1:
        return make_id()
    addr = addr or os.environ.get("MASTER_ADDR", "127.0.0.1")
    port = int(port if port is not None else int(os.environ.get("MASTER_PORT", "29500")) + 1)
    token = _id_token()
    if rank == 0:
        uid = make_id()
        served = set()
        deadline = time.monotonic() + timeout
        with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as srv:
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind((addr, port))
            srv.listen(world)
            while len(served) < world - 1:
                left = deadline - time.monotonic()
                if left <= 0:
                    raise TimeoutError(f"group id exchange: ranks {sorted(set(range(1, world)) - served)} never asked")
                srv.settimeout(left)
                try:
                    conn, _peer = srv.accept()
                except socket.timeout:
                    raise TimeoutError(f"group id exchange: ranks {sorted(set(range(1, world)) - served)} never asked") from None
                with conn:
                    try:
                        conn.settimeout(5.0)
                        r, n = struct.unpack("<II", _recv_exact(conn, 8))
                        if n > 64 or _recv_exact(conn, n) != token or not 1 <= r < world:
                            continue  # not one of ours
                        conn.sendall(struct.pack("<I", len(uid)) + uid)
                        served.add(r)
                    except (OSError, ConnectionError, struct.error):
                        continue
            # every rank has been answered once; an answer may still have been lost on the way, so the port stays
            # open for a short grace period and a rank that asks again is answered again
            # (ONE deadline for the whole grace period: a client that keeps reconnecting cannot hold rank 0 back)
            grace_deadline = time.monotonic() + grace
            while True:
                left = grace_deadline - time.monotonic()
                if left <= 0:
                    break
                srv.settimeout(left)
                try:
                    conn, _peer = srv.accept()
                except (socket.timeout, OSError):
                    break
                with conn:
                    try:
                        conn.settimeout(max(0.05, min(2.0, grace_deadline - time.monotonic())))
                        r, n = struct.unpack("<II", _recv_exact(conn, 8))
                        if n <= 64 and _recv_exact(conn, n) == token and 1 <= r < world:
                            conn.sendall(struct.pack("<I", len(uid)) + uid)
                    except (OSError, ConnectionError, struct.error):
                        continue
        return uid
    deadline = time.monotonic() + timeout
    while True:
        try:
            with socket.create_connection((addr, port), timeout=5.0) as conn:
                conn.sendall(struct.pack("<II", int(rank), len(token)) + token)
                head = _recv_exact(conn, 4)
                return _recv_exact(conn, struct.unpack("<I", head)[0])
        except (ConnectionError, socket.timeout, OSError):
            if time.monotonic() > deadline:
                raise
            time.sleep(0.05)


def _recv_exact(conn, n):
    out = b""
    while len(out) < n:
        chunk = conn.recv(n - len(out))
        if not chunk:
            raise ConnectionError("peer closed the connection while sending the group id")
        out += chunk
    return out


def shard_range(m, rank, world):
    """Contiguous range [lo, hi) of ``rank``: global row order is preserved across ranks (the same
    rule as ``gpso_shard_range``)."""
    base, extra = divmod(int(m), int(world))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


# -- collective calls: native for HIP engines ---------------------------------------------------------
def broadcast_posterior(engine, src=0):
    """Make the posterior resident on rank ``src`` resident on every rank of the engine's group."""
    engine.broadcast_posterior(src)


def best_ucb_sharded(engine, local_leaves, m_global, varsigma, seg_off=None):
    """Score this rank's leaf shard (rows ``shard_range(m_global, rank, world)`` of the batch) and agree
    on the global winner of every segment.  Returns (idx, mean, var, ucb) arrays of length nseg --
    identical on every rank, and identical to ``engine.best_ucb(all_leaves, varsigma, seg_off)``."""
    return engine.best_ucb_sharded(local_leaves, m_global, varsigma, seg_off)


def best_ucb_grow_sharded(engine, bounds, depth, varsigma):
    return engine.best_ucb_grow_sharded(bounds, depth, varsigma)


# -- ONE process, several GPUs --------------------------------------------------------------------------
class HipGPEngineGroup:
    """``HipGPEngine`` interface over several devices of one process: the fit runs on the first device,
    every predict-type call is sharded over all of them.  Each engine is driven by its own thread (the
    C-ABI calls release the GIL; the RCCL collectives inside them need all ranks in flight at once)."""

    def __init__(self, dtype="float64", devices=(0,), engine_cls=None, make_id=None, posterior="broadcast",
                 **engine_options):
        """``posterior``: how the peers get the fitted posterior.  "broadcast" (default): the first device fits, ONE
        ncclBroadcast moves the predict-ready range of its posterior arena to the others.  "replicate": every device
        runs the same (bit-deterministic) fit on its own copy of the data -- no bulk collective at all; the ranks'
        posterior fingerprints (``gpso_posterior_hash``) must agree before the first predict-type call on a new
        posterior.  ``engine_cls`` / ``make_id``: test hooks (the per-device engine class, default ``HipGPEngine``,
        and the source of the group id, default ``unique_id``)."""
        if posterior not in ("broadcast", "replicate"):
            raise ValueError("posterior must be 'broadcast' or 'replicate'")
        self.posterior = posterior
        if engine_cls is None:
            from .engine import HipGPEngine as engine_cls

        self.devices = [int(dev) for dev in devices]
        if not self.devices:
            raise ValueError("devices must name at least one GPU")
        self.world = len(self.devices)
        self.engines = [engine_cls(dtype, device=dev, **engine_options) for dev in self.devices]
        self.dtype_name, self.dtype = self.engines[0].dtype_name, self.engines[0].dtype
        self.device = self.devices[0]
        self._pool = ThreadPoolExecutor(self.world)
        self.abort_after = 30.0  # seconds a failed group call waits for the other ranks before aborting them
        self._broken = False     # a communicator was aborted: the group cannot make group calls any more
        uid = (make_id or unique_id)()
        self._all(lambda r, e: e.comm_init(r, self.world, uid), collective=True)
        self._stale = False  # peers lag behind the root's posterior?
        self._stale_rows = False     # ... by appended rows only (gpso_broadcast_posterior_rows applies)
        self._stale_rows_ok = False  # the peers hold the posterior the next append extends
        self.last_handoff = None     # ("rows" | "whole", bytes) of the last hand-off
        self.n = self.d = 0

    def _all(self, fn, collective=False):
        """fn(rank, engine) on every engine, each on its own thread; returns the results in rank order (the
        first exception, if any, is raised after all threads are back).  ``collective``: fn contains RCCL
        collectives (comm_init, broadcast_posterior, the *_sharded calls).  The library keeps every rank inside
        every collective of a call whatever fails locally, so the threads come back together; should one
        raise while others are still inside a collective after ``abort_after`` seconds (a HIP / RCCL failure
        in the middle of a call), their communicators are aborted (``gpso_comm_abort``) so that they return
        an error too instead of waiting for ever -- the group is then broken (``_check_usable`` refuses further
        group calls; ``rebuild()`` makes new communicators).  Fan-outs without a collective (predict shards,
        options) have nobody to wait for: a slow rank is simply joined."""
        if collective:
            self._check_usable()
        futs = [self._pool.submit(fn, r, e) for r, e in enumerate(self.engines)]
        done, pending = wait(futs, return_when=FIRST_EXCEPTION)
        if collective and pending and any(f.exception() is not None for f in done):
            done2, pending = wait(pending, timeout=self.abort_after)
            for f in pending:
                eng = self.engines[futs.index(f)]
                if hasattr(eng, "comm_abort"):
                    eng.comm_abort()
                self._broken = True
        wait(futs)
        out, err = [], None
        for f in futs:
            try:
                out.append(f.result())
            except Exception as exc:  # noqa: BLE001 - re-raised below
                err = err or exc
                out.append(None)
        if err is not None:
            raise err
        return out

    def _check_usable(self):
        if self._broken:
            raise L.GpsoHipError(L.E_STATE, "a communicator of this group was aborted after a failed group call: "
                                            "HipGPEngineGroup.rebuild() makes new ones (or create a new group)")

    def rebuild(self, make_id=None):
        """New communicators on every engine after an abort (``gpso_comm_destroy`` + ``gpso_comm_init``); the
        posterior is broadcast again by the next predict-type call."""
        for e in self.engines:
            e.comm_destroy()
        self._broken = False
        uid = (make_id or unique_id)()
        self._all(lambda r, e: e.comm_init(r, self.world, uid), collective=True)
        self._new_posterior()

    def close(self):
        for e in getattr(self, "engines", []):
            e.close()
        if getattr(self, "_pool", None) is not None:
            self._pool.shutdown(wait=True)
            self._pool = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # fit: first device only ("broadcast"), or the same fit on every device ("replicate")
    def _fitters(self, fn):
        if self.posterior == "replicate" and self.world > 1:
            return self._all(lambda r, e: fn(e))[0]
        return fn(self.engines[0])

    def _new_posterior(self):
        self._stale = True
        self._stale_rows = False
        self._stale_rows_ok = False

    def set_data(self, X, y):
        self._fitters(lambda e: e.set_data(X, y))
        self.n, self.d = self.engines[0].n, self.engines[0].d
        self._new_posterior()

    def fit_eval(self, *a, **kw):
        self._new_posterior()
        return self._fitters(lambda e: e.fit_eval(*a, **kw))

    def fit_eval_u(self, *a, **kw):
        self._new_posterior()
        return self._fitters(lambda e: e.fit_eval_u(*a, **kw))

    def append(self, Xnew, ynew):
        """``HipGPEngine.append`` on the fitting rank(s); the peers get the extended posterior like a fitted one."""
        out = self._fitters(lambda e: e.append(Xnew, ynew))
        self.n = self.engines[0].n
        # "broadcast" groups: the next predict-type call moves only what the append wrote (gpso_broadcast_posterior_rows; the
        # library falls back to the whole range on every rank when a peer does not hold the base) -- not the whole posterior
        self._stale = True
        self._stale_rows = self._stale_rows_ok and out[1]
        return out

    def set_posterior(self, *a, **kw):
        self._fitters(lambda e: e.set_posterior(*a, **kw))
        self.n, self.d = self.engines[0].n, self.engines[0].d
        self._new_posterior()

    def _sync_posterior(self):
        if self._stale and self.world > 1:
            if self.posterior == "replicate":
                hashes = self._all(lambda r, e: e.posterior_hash())
                if len(set(hashes)) != 1:
                    raise L.GpsoHipError(L.E_STATE, "replicated fits disagree: posterior fingerprints "
                                         + ", ".join(f"{h:016x}" for h in hashes))
            elif self._stale_rows and hasattr(self.engines[0], "broadcast_posterior_rows"):
                rows = self._all(lambda r, e: e.broadcast_posterior_rows(0), collective=True)
                self.last_handoff = ("rows" if rows[0] else "whole", self.engines[0].last_count(0))
            else:
                self._all(lambda r, e: e.broadcast_posterior(0), collective=True)
                self.last_handoff = ("whole", self.engines[0].last_count(0) if hasattr(self.engines[0], "last_count") else None)
        self._stale = False
        self._stale_rows = False
        self._stale_rows_ok = True  # (from here on an append alone can be handed on by rows)

    # predict-type calls: sharded
    def predict(self, xs, out=None):
        if out is not None:
            raise ValueError("device outputs are per-engine: use the engines of the group directly")
        self._sync_posterior()
        xs = np.asarray(xs)
        m = xs.shape[0]
        parts = self._all(lambda r, e: e.predict(xs[slice(*shard_range(m, r, self.world))]))
        return np.concatenate([p[0] for p in parts]), np.concatenate([p[1] for p in parts])

    def best_ucb(self, xs, varsigma, seg_off=None):
        self._sync_posterior()
        xs = np.asarray(xs)
        m = xs.shape[0]
        res = self._all(lambda r, e: e.best_ucb_sharded(xs[slice(*shard_range(m, r, self.world))], m, varsigma,
                                                        seg_off), collective=True)
        return res[0]

    def best_ucb_grow(self, bounds, depth, varsigma):
        self._sync_posterior()
        return self._all(lambda r, e: e.best_ucb_grow_sharded(bounds, depth, varsigma), collective=True)[0]

    # options: the arithmetic options of a group must match on every rank (the library checks dtype and
    # predict math at the broadcast; generation, self-test and tolerances would silently differ otherwise)
    def set_predict_math(self, mode):
        self._all(lambda r, e: e.set_predict_math(mode))
        self._new_posterior()

    def set_generation(self, mode):
        self._all(lambda r, e: e.set_generation(mode))
        self._new_posterior()

    def set_precision_check(self, on):
        self._all(lambda r, e: e.set_precision_check(on))

    def set_timing(self, on):
        self._all(lambda r, e: e.set_timing(on))

    def set_tolerances(self, tol_var=None, tol_mean=None):
        self._all(lambda r, e: e.set_tolerances(tol_var, tol_mean))

    def synchronize(self):
        self._all(lambda r, e: e.synchronize())

    # read-only introspection of the fitting rank (first device); anything that would CHANGE one engine only is
    # not forwarded
    _ROOT_READS = ("precision_info", "last_ms", "last_count", "fit_math", "padded_n", "get_matrix", "get_vector", "grow",
                   "grow_rows", "posterior_buffers", "rank")

    def __getattr__(self, name):
        if name in HipGPEngineGroup._ROOT_READS:
            return getattr(self.engines[0], name)
        raise AttributeError(f"HipGPEngineGroup has no attribute {name!r} (per-engine calls: use .engines[i])")

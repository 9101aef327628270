"""
GP surrogate of the objective surface -- the drop-in boundary of the hot path.

Mirrors the reference's operator interface (names, argument meaning, error behaviour):

* ``GPPoint`` .............. gpso/gp_surrogate.py:24-36
* ``GPListOfPoints`` ....... gpso/gp_surrogate.py:39-118  (1e-12 L2 duplicate rule)
* ``GPSurrogate`` .......... gpso/gp_surrogate.py:121-385 (append / gp_predict / gp_eval_best_ucb /
                             gp_update / properties)
* ``GPRSurrogate`` ......... gpso/gp_surrogate.py:388-533 (exact GP regression; ``_gp_train``)

What differs is underneath: ``gpflow_model`` is a ``HipGPR`` whose training loss, gradient and
``predict_y`` run as HIP kernels on the MI355X (pygpso_amd/csrc), the point store answers the
duplicate queries with one vectorised pass over a coordinate array instead of a Python scan, and
``gp_eval_best_ucb_grow`` scores whole ternary sub-trees without materialising them on the host.
``VGPSurrogate`` (variational GP) is out of scope.
"""
from __future__ import annotations

import json
import logging
import os
from collections import namedtuple

import math

import numpy as np
from scipy.special import erfcinv

from .kernels import KERNEL_CLASSES, Constant, Kernel, Matern52, MeanFunction, Scipy, Zero
from .model import HipGPR
from .utils import JSON_EXT, PointLabels

DUPLICATE_TOLERANCE = 1.0e-12
NORM_PARAMS_BOUNDS = (0, 1)



class GPPoint(namedtuple("GPPoint", ["normed_coord", "score_mu", "score_sigma", "score_ucb", "label"])):
    """One stored point: normalised coordinates, mean score, VARIANCE (the field is called
    score_sigma in the reference but holds the variance, gpso/gp_surrogate.py:305), UCB, label.
    Equality compares the coordinate arrays element-wise."""

    __slots__ = ()

    def __eq__(self, other):
        if not isinstance(other, tuple) or len(other) != len(self):
            return False
        return bool(np.array_equal(self[0], other[0])) and tuple(self[1:]) == tuple(other[1:])

    def __ne__(self, other):
        return not self.__eq__(other)

    __hash__ = None


_NONE = []


class GPListOfPoints(list):
    """List of ``GPPoint`` whose ``append`` de-duplicates by coordinates.

    Same observable behaviour as the reference: a new point within 1e-12 (L2) of stored points
    replaces every such point that is not ``evaluated`` and is not appended; the constructor does
    not de-duplicate.  ``append`` additionally RETURNS the index the point now lives at (first
    duplicate, or the new last position)."""

    # The index hashes a PROJECTION w.x of the coordinates, bucket width 1e-6: two points closer than the duplicate
    # tolerance (1e-12, L2) differ by at most |w| x 1e-12 < 1.5e-12 in it, so a duplicate of x can only sit in the bucket of
    # w.x or one of its two neighbours.  (Rounds 1-5 hashed coordinate 0 alone: the centres of a ternary tree share their
    # first coordinate -- in D = 12 the tree splits other dimensions for a long time --, nearly every point fell into ONE
    # bucket and a look-up scanned the whole list in Python: 250 us per look-up at 1 000 points, O(P^2) per gp_update.  With
    # w_i = 1 / sqrt(i-th prime) distinct grid points have distinct projections.)
    _BUCKET = 1.0e-6
    _PRIMES = (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37, 41, 43, 47, 53, 59, 61, 67, 71, 73, 79, 83, 89, 97, 101, 103, 107,
               109, 113, 127, 131, 137, 139, 149, 151, 157, 163, 167, 173, 179, 181, 191, 193, 197, 199, 211, 223, 227, 229,
               233, 239, 241, 251, 257, 263, 269, 271, 277, 281, 283, 293, 307, 311)
    _W = tuple(1.0 / math.sqrt(p) for p in _PRIMES)  # |w|^2 = sum 1 / p < 2.1 over these 64

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        assert all(isinstance(p, GPPoint) for p in self)
        self._rows = []      # coordinates of self[i] as a tuple of Python floats
        self._dirty = True
        self._buckets = {}   # bucket of the projection -> indices (ascending)
        # every point came in through append(): no two stored points lie within the duplicate tolerance of each other (the
        # constructor does not de-duplicate, list surgery may break it: cleared by anything but append / score updates)
        self._unique = len(self) == 0

    # -- index kept in sync ------------------------------------------------------------------------
    # (plain Python floats on purpose: a look-up touches one or two candidate rows of D numbers, and the
    # fixed cost of any numpy call is larger than that whole computation)
    def _index(self):
        if self._dirty or len(self._rows) != len(self):
            self._rows = [tuple(np.asarray(p.normed_coord, dtype=np.float64).reshape(-1).tolist()) for p in self]
            self._buckets = {}
            for i, r in enumerate(self._rows):
                self._buckets.setdefault(self._bucket_of(r), []).append(i)
            self._dirty = False

    @classmethod
    def _bucket_of(cls, row):
        """bucket of the projection w.row (dimensions beyond the 64 weights wrap around: still a valid projection)"""
        w = cls._W
        if len(row) <= len(w):
            proj = sum(a * b for a, b in zip(row, w))
        else:
            proj = sum(a * w[i % len(w)] for i, a in enumerate(row))
        return int(math.floor(proj / cls._BUCKET))

    def _matches(self, coords):
        """Indices (ascending) of the stored points within the duplicate tolerance of ``coords`` --
        the reference's linear scan (gpso/gp_surrogate.py:68-101), answered from a hash on the first
        coordinate: candidates come from three buckets, the exact distance test decides."""
        self._index()
        if not self._rows:
            return []
        c = np.asarray(coords, dtype=np.float64).reshape(-1).tolist()
        b = self._bucket_of(c)
        get = self._buckets.get
        cand = get(b - 1, _NONE) + get(b, _NONE) + get(b + 1, _NONE)
        if not cand:
            return []
        if len(cand) > 1:
            cand = sorted(cand)
        rows, hits = self._rows, []
        for i in cand:
            d2 = 0.0
            for a, x in zip(rows[i], c):
                t = a - x
                d2 += t * t
            if math.sqrt(d2) < DUPLICATE_TOLERANCE:
                hits.append(i)
        return hits

    def __setitem__(self, idx, value):
        super().__setitem__(idx, value)
        self._unique = False  # (arbitrary surgery: the store can no longer vouch for uniqueness; append() keeps it itself)
        n = len(self._rows)
        if isinstance(idx, int) and not self._dirty and n == len(self) and -n <= idx < n:
            i = idx % n
            row = tuple(np.asarray(value.normed_coord, dtype=np.float64).reshape(-1).tolist())
            old_b, new_b = self._bucket_of(self._rows[i]), self._bucket_of(row)
            self._rows[i] = row
            if old_b != new_b:
                self._buckets[old_b].remove(i)
                self._buckets.setdefault(new_b, []).append(i)
        else:
            self._dirty = True

    # -- reference API ----------------------------------------------------------------------------
    def append(self, point):
        assert isinstance(point, GPPoint)
        hits = self._matches(point.normed_coord)
        if hits:
            unique = self._unique
            for i in hits:
                if self[i].label != PointLabels.evaluated:
                    self[i] = point
            self._unique = unique  # (a point replaced by one within the tolerance of it: uniqueness is kept)
            return hits[0]
        n = len(self)
        super().append(point)
        if not self._dirty and len(self._rows) == n:
            row = tuple(np.asarray(point.normed_coord, dtype=np.float64).reshape(-1).tolist())
            self._rows.append(row)
            self._buckets.setdefault(self._bucket_of(row), []).append(n)
        else:
            self._dirty = True
        return n

    def update_scores(self, indices, mean, var, varsigma):
        """New (mean, var, ucb) for the points at ``indices``, coordinates and labels kept -- what re-appending a point with
        the SAME coordinates does (it overwrites its own entry), without the look-up.  Only valid while no two stored points
        are duplicates of each other (``_unique``); returns False otherwise and changes nothing."""
        if not self._unique:
            return False
        setitem = list.__setitem__
        for i, m, v in zip(indices, mean, var):
            p = self[i]
            m, v = float(m), float(v)
            setitem(self, i, GPPoint(p.normed_coord, m, v, float(m + varsigma * v), p.label))  # (coordinates unchanged: the index stays valid)
        return True

    def find_index_by_coords(self, coords):
        hits = self._matches(coords)
        return hits[0] if hits else None

    def find_by_coords(self, coords):
        i = self.find_index_by_coords(coords)
        return None if i is None else self[i]

    # -- persistence (same JSON schema as the reference, gpso/gp_surrogate.py:103-118) ------------
    def save(self, filename):
        if not filename.endswith(JSON_EXT):
            filename += JSON_EXT
        rows = []
        for p in self:
            rows.append({
                "normed_coord": np.asarray(p.normed_coord).tolist(),
                "score_mu": float(p.score_mu),
                "score_sigma": float(p.score_sigma),
                "score_ucb": float(p.score_ucb),
                "label": p.label.name,
            })
        with open(filename, "w") as fh:
            fh.write(json.dumps(rows))

    @classmethod
    def from_file(cls, filename):
        if not filename.endswith(JSON_EXT):
            filename += JSON_EXT
        with open(filename) as fh:
            rows = json.load(fh)
        return cls([
            GPPoint(np.array(r["normed_coord"]), r["score_mu"], r["score_sigma"], r["score_ucb"],
                    PointLabels[r["label"]])
            for r in rows
        ])


def _invalidating(name):
    base = getattr(list, name)

    def method(self, *args, **kwargs):
        self._dirty = True
        self._unique = False
        return base(self, *args, **kwargs)

    method.__name__ = name
    return method


for _name in ("extend", "insert", "pop", "remove", "clear", "sort", "reverse", "__delitem__", "__iadd__"):
    setattr(GPListOfPoints, _name, _invalidating(_name))
del _name


class GPSurrogate:
    """Base class: point bookkeeping + predict/UCB on top of ``self.gpflow_model``."""

    POINTS_FILE = f"points{JSON_EXT}"
    GPR_FILE = f"GPRmodel{JSON_EXT}"
    GPR_INFO = f"GPRinfo{JSON_EXT}"

    @classmethod
    def from_saved(cls, folder):
        raise NotImplementedError

    def __init__(self, gp_kernel, gp_meanf=None, optimiser=None, varsigma=erfcinv(0.01), points=None,
                 gpflow_model=None, dtype="float64", device=0, engine_options=None, devices=None):
        """
        :param gp_kernel: kernel spec (``pygpso_amd.kernels.Matern52(...)`` etc.)
        :param gp_meanf: mean-function spec (``Constant(c)``) or None
        :param optimiser: object with ``minimize(closure, variables)``; default ``Scipy()`` (L-BFGS-B)
        :param varsigma: UCB = mean + varsigma * VAR (gpso/gp_surrogate.py:150-155,326)
        :param points: initial list of ``GPPoint``
        :param gpflow_model: an initialised ``HipGPR`` (used when loading a saved surrogate)
        :param dtype: arithmetic of the device kernels: "float64" (the reference's; parity), "mixed"
            (fit in float64, predictions in float32 / split bf16) or "float32" (everything in float32).
            Float predictions are guarded by a self-test; a model whose posterior fails it moves to the
            next more precise arithmetic on the device by itself (``HipGPR``).
        :param device: HIP device index
        :param devices: several HIP device indices: the fit runs on the first, every leaf-UCB / predict
            call is sharded over all of them (RCCL behind the C-ABI, ``HipGPEngineGroup``); results are
            bit-identical to a single device's
        :param engine_options: extra ``HipGPEngine`` keyword arguments (predict_math, generation, ...)
        """
        self.gpflow_model = gpflow_model
        self.gp_varsigma = float(varsigma)
        assert isinstance(gp_kernel, Kernel)
        self.gp_kernel = gp_kernel
        assert gp_meanf is None or isinstance(gp_meanf, MeanFunction)
        self.gp_meanf = gp_meanf
        optimiser = optimiser if optimiser is not None else Scipy()
        assert hasattr(optimiser, "minimize")
        self.optimiser = optimiser
        self.dtype = dtype
        self.devices = [int(v) for v in devices] if devices is not None else None
        self.device = self.devices[0] if self.devices else device
        self.engine_options = dict(engine_options or {})
        self.points = GPListOfPoints(points or list())

    # -- bookkeeping properties (gpso/gp_surrogate.py:174-257) -----------------------------------
    def _with_label(self, label):
        return [p for p in self.points if p.label == label]

    @property
    def num_evaluated(self):
        return len(self._with_label(PointLabels.evaluated))

    @property
    def num_gp_based(self):
        return len(self._with_label(PointLabels.gp_based))

    @property
    def highest_score(self):
        cand = self._with_label(PointLabels.evaluated)
        if cand:
            return max(cand, key=lambda p: p.score_mu)  # max() keeps the first of equal maxima

    @property
    def highest_ucb(self):
        cand = self._with_label(PointLabels.gp_based)
        if cand:
            return max(cand, key=lambda p: p.score_ucb)

    @property
    def current_training_data(self):
        ev = self._with_label(PointLabels.evaluated)
        return np.array([p.normed_coord for p in ev]), np.array([p.score_mu for p in ev])

    @property
    def gp_based_coords(self):
        return np.array([p.normed_coord for p in self._with_label(PointLabels.gp_based)])

    # -- training-data intake ----------------------------------------------------------------------
    def _gp_train(self, x, y):
        raise NotImplementedError

    def append(self, coords, scores):
        """Store evaluated points (normalised coordinates [n, D], scores [n])."""
        assert coords.ndim == 2
        assert scores.ndim == 1
        assert coords.shape[0] == scores.shape[0]
        for c, s in zip(coords, scores):
            self.points.append(GPPoint(c, s, 0.0, 0.0, PointLabels.evaluated))

    # -- predict / UCB ------------------------------------------------------------------------------
    def _require_model(self):
        assert isinstance(self.gpflow_model, HipGPR), "GP model not trained yet"

    def gp_predict(self, normed_coords):
        """predict_y at ``normed_coords`` and store every row as a gp_based point."""
        self._require_model()
        mean, var = self.gpflow_model.predict_y(normed_coords)
        for i in range(normed_coords.shape[0]):
            m, v = float(mean[i, 0]), float(var[i, 0])
            self.points.append(GPPoint(normed_coords[i, :], m, v, float(m + self.gp_varsigma * v),
                                       PointLabels.gp_based))

    def gp_eval_best_ucb(self, normed_coords):
        """(mean, var, ucb) of the row with the highest ucb = mean + varsigma * var; stores nothing."""
        self._require_model()
        _, mean, var, ucb = self.gpflow_model.best_ucb(normed_coords, self.gp_varsigma)
        return float(mean[0]), float(var[0]), float(ucb[0])

    def gp_eval_best_ucb_grow(self, leaf_bounds, depth):
        """MI355X path of ``gp_eval_best_ucb(leaf.grow(depth))`` for several leaves at once: the
        ternary sub-tree centres are generated on the device (bit-identical to ``LeafNode.grow``)
        and never cross PCIe.  ``leaf_bounds``: [nleaf, D, 2].  Returns a list of (mean, var, ucb)."""
        self._require_model()
        _, mean, var, ucb = self.gpflow_model.best_ucb_grow(np.asarray(leaf_bounds, dtype=np.float64),
                                                             depth, self.gp_varsigma)
        return [(float(m), float(v), float(u)) for m, v, u in zip(mean, var, ucb)]

    def gp_update(self):
        """Retrain on the evaluated points, then re-predict every gp_based point."""
        x_train, y_train = self.current_training_data
        if logging.getLogger().isEnabledFor(logging.DEBUG):  # (formatting the arrays is not free)
            logging.debug(f"Retraining GPR with x data: {x_train}; y data: {y_train}")
        self._gp_train(x=x_train, y=y_train[:, np.newaxis])
        # re-predict every gp-based point.  The reference re-appends them (gpso/gp_surrogate.py:341-342): each overwrites its
        # own entry -- done here by index, one predict call and no look-ups, while the store can vouch that no two points are
        # duplicates of each other; otherwise the reference's way
        idx = [i for i, p in enumerate(self.points) if p.label == PointLabels.gp_based]
        if idx:
            coords = np.array([self.points[i].normed_coord for i in idx])
            if getattr(self.points, "_unique", False):
                mean, var = self.gpflow_model.predict_y(coords)
                if self.points.update_scores(idx, np.asarray(mean)[:, 0], np.asarray(var)[:, 0], self.gp_varsigma):
                    return
            self.gp_predict(coords)

    def save(self, folder):
        raise NotImplementedError


class GPRSurrogate(GPSurrogate):
    """Exact GP regression surrogate (the reference's default)."""

    # hook for tests / multi-GPU: a callable returning the engine object HipGPR should drive
    # (default None -> a HipGPEngine on ``device``; there is no CPU engine in this package)
    engine_factory = None

    def __init__(self, gp_kernel, gp_meanf=None, optimiser=None, varsigma=erfcinv(0.01),
                 gauss_likelihood_sigma=1.0e-3, points=None, gpflow_model=None, dtype="float64",
                 device=0, engine_options=None, devices=None, refit_every=1, refit_guard=2.0):
        """
        :param gauss_likelihood_sigma: initial noise VARIANCE of the Gaussian likelihood (the
            reference passes it as ``noise_variance`` despite the name, gpso/gp_surrogate.py:494)
        :param refit_every: 1 (default): every ``gp_update`` re-optimises the hyper-parameters, as the reference does
            (gpso/gp_surrogate.py:496-503).  c > 1 (opt-in, NOT the reference's behaviour): only every c-th update
            re-optimises; the updates in between keep the hyper-parameters and extend the device posterior by the new
            points in place (``gpso_append``: O(N^2 k) instead of 10-45 O(N^3) loss evaluations)
        :param refit_guard: (with ``refit_every`` > 1) how far the kept hyper-parameters may drift before an appended
            posterior counts as stale: an append's NLML increment per new point is -log p(y_new | data, theta); when it
            exceeds the fit's own average (NLML / N) by more than ``refit_guard`` nats per point the new points are
            surprising under theta and THIS update re-optimises instead of waiting for the c-th.  None: no guard
        """
        super().__init__(gp_kernel=gp_kernel, gp_meanf=gp_meanf, optimiser=optimiser,
                         varsigma=varsigma, points=points, gpflow_model=gpflow_model, dtype=dtype,
                         device=device, engine_options=engine_options, devices=devices)
        self.gp_lik_sigma = gauss_likelihood_sigma
        self.refit_every = max(1, int(refit_every))
        self.refit_guard = None if refit_guard is None else float(refit_guard)
        self.guard_refits = 0  # updates the guard turned into re-optimisations
        self._updates = 0  # gp_update calls so far (refit_every counts them)

    @classmethod
    def default(cls, dtype="float64", device=0, engine_options=None, devices=None):
        """Matern-5/2 (l = 0.25, s2 = 1), constant mean 0, L-BFGS-B, noise 1e-3
        (gpso/gp_surrogate.py:418-434)."""
        return cls(
            gp_kernel=Matern52(lengthscales=np.sum(NORM_PARAMS_BOUNDS) * 0.25, variance=1.0),
            gp_meanf=Constant(0.0),
            optimiser=Scipy(),
            varsigma=erfcinv(0.01),
            gauss_likelihood_sigma=1.0e-3,
            dtype=dtype,
            device=device,
            engine_options=engine_options,
            devices=devices,
        )

    def _gp_train(self, x, y):
        assert x.shape[0] == y.shape[0]
        assert x.ndim == 2 and y.ndim == 2
        if self.gpflow_model is None:
            engine = self.engine_factory() if self.engine_factory is not None else None
            self.gpflow_model = HipGPR(data=(x, y), kernel=self.gp_kernel, mean_function=self.gp_meanf,
                                       noise_variance=self.gp_lik_sigma, dtype=self.dtype,
                                       device=self.device, engine=engine,
                                       engine_options=self.engine_options, devices=self.devices)
        else:
            n_old = self.gpflow_model.data[0].shape[0]
            new_rows = None
            if self.refit_every > 1 and self._updates % self.refit_every != 0 and x.shape[0] > n_old:
                new_rows = self._rows_beyond(self.gpflow_model.data, x, y)
            if new_rows is not None:
                # the model's points are all still there with their scores: extend the posterior at the kept
                # hyper-parameters by the others.  (The evaluated points do NOT only grow at the end of the list: an
                # evaluation of a point that was stored gp-based overwrites that entry in place, gpso/gp_surrogate.py:
                # 68-101 -- the model then holds the points in ITS order of arrival, a permutation of the list's; the next
                # re-optimisation sets the list's order again.)
                try:
                    model = self.gpflow_model
                    model._ensure_resident()
                    nlml_before = float(model._last_nlml)
                    model.append_data(x[new_rows], y[new_rows])
                    per_new = (float(model._last_nlml) - nlml_before) / (x.shape[0] - n_old)
                    if self.refit_guard is None or not (per_new > nlml_before / n_old + self.refit_guard):
                        self._updates += 1
                        return
                    # the new points are much less likely under the kept hyper-parameters than the old ones were on
                    # average: theta has drifted -- re-optimise now (the model already holds all points)
                    self.guard_refits += 1
                    logging.info(f"appended points cost {per_new:.2f} nats each against {nlml_before / n_old:.2f} on average: "
                                 "re-optimising the hyper-parameters on this update")
                except np.linalg.LinAlgError as err:
                    # the appended block is not positive definite at the kept hyper-parameters (in the engine's
                    # arithmetic): the model holds the N + k points with no posterior -- this update re-optimises instead
                    logging.warning(f"{err}; this update re-optimises the hyper-parameters instead of appending")
            else:
                self.gpflow_model.data = (x, y)  # hyper-parameters warm-start from the last optimum
        self._updates += 1
        self.optimiser.minimize(self.gpflow_model.training_loss, self.gpflow_model.trainable_variables)

    @staticmethod
    def _rows_beyond(data, x, y):
        """Indices of the rows of (x, y) that the model's data does not hold, when EVERY row of the model's data is among
        (x, y) bit for bit with its score (any order) -- else None: the update cannot be an append."""
        old_x, old_y = data
        if old_x.shape[1] != x.shape[1] or x.shape[0] <= old_x.shape[0]:
            return None
        rec = np.dtype([("", np.float64)] * (x.shape[1] + 1))
        new_v = np.ascontiguousarray(np.hstack([x, y.reshape(-1, 1)]), dtype=np.float64).view(rec).ravel()
        old_v = np.ascontiguousarray(np.hstack([old_x, old_y.reshape(-1, 1)]), dtype=np.float64).view(rec).ravel()
        held = np.isin(new_v, old_v)
        if int(held.sum()) != old_v.shape[0] or np.unique(old_v).shape[0] != old_v.shape[0]:
            return None
        return np.flatnonzero(~held)

    # -- persistence: points JSON (reference schema) + hyper-parameters as plain JSON ---------------
    def save(self, folder):
        os.makedirs(folder, exist_ok=True)
        self.points.save(os.path.join(folder, self.POINTS_FILE))
        model = self.gpflow_model
        params = {k: np.asarray(v).tolist() for k, v in model.parameter_dict().items()}
        with open(os.path.join(folder, self.GPR_FILE), "w") as fh:
            fh.write(json.dumps(params))
        info = {
            "gpr_kernel": model.kernel.name,
            "gpr_kernel_shape": list(np.shape(model.kernel.lengthscales)),
            "gpr_meanf": type(model.mean_function).__name__,
            "gpr_meanf_shape": [],
            "gp_varsigma": self.gp_varsigma,
            "gp_likelihood": self.gp_lik_sigma,
            "optimiser": [type(self.optimiser).__name__],
            "dtype": self.dtype,
            "refit_every": self.refit_every,  # (not in the reference's schema: an extra key; 1 = the reference's behaviour)
            "refit_guard": self.refit_guard,
        }
        with open(os.path.join(folder, self.GPR_INFO), "w") as fh:
            fh.write(json.dumps(info))

    @classmethod
    def from_saved(cls, folder, device=0, devices=None):
        points = GPListOfPoints.from_file(os.path.join(folder, cls.POINTS_FILE))
        ev = [p for p in points if p.label == PointLabels.evaluated]
        x = np.array([p.normed_coord for p in ev])
        y = np.array([p.score_mu for p in ev])[:, np.newaxis]
        with open(os.path.join(folder, cls.GPR_INFO)) as fh:
            info = json.load(fh)
        with open(os.path.join(folder, cls.GPR_FILE)) as fh:
            params = json.load(fh)
        assert info["gpr_kernel"] in KERNEL_CLASSES
        kernel = KERNEL_CLASSES[info["gpr_kernel"]](
            lengthscales=np.array(params[".kernel.lengthscales"]), variance=params[".kernel.variance"])
        meanf = Constant(params[".mean_function.c"]) if info["gpr_meanf"] == "Constant" else Zero()
        assert info["optimiser"][0] == "Scipy", f"{info['optimiser']} not currently supported."
        engine = cls.engine_factory() if cls.engine_factory is not None else None
        model = HipGPR(data=(x, y), kernel=kernel, mean_function=meanf,
                       noise_variance=params[".likelihood.variance"], dtype=info.get("dtype", "float64"),
                       device=device, engine=engine, devices=devices)
        return cls(gp_kernel=kernel, gp_meanf=meanf, optimiser=Scipy(),
                   gauss_likelihood_sigma=info["gp_likelihood"], varsigma=info["gp_varsigma"],
                   points=points, gpflow_model=model, dtype=info.get("dtype", "float64"), device=device,
                   devices=devices, refit_every=info.get("refit_every", 1), refit_guard=info.get("refit_guard", 2.0))

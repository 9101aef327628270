"""
``HipGPR``: the object the surrogate keeps in ``.gpflow_model``.

Stands where ``gpflow.models.GPR`` stands in the reference (gpso/gp_surrogate.py:488-503,
298, 325; gpso/plotting.py:351-356 reaches it through ``gp_surr.gpflow_model.predict_y``):
holds data + hyper-parameters, evaluates ``training_loss`` and answers ``predict_y`` -- by calling
the HIP engine.  The parameter transforms are GPflow-2's: softplus for
lengthscales / kernel variance, 1e-6 + softplus for the likelihood variance, identity for the mean.
"""
from __future__ import annotations

import logging
import math
import types

import numpy as np

from . import _lib as L
from .engine import HipGPEngine
from .kernels import Constant, Kernel, MeanFunction, Zero

NOISE_FLOOR = 1.0e-6  # gpflow.likelihoods.Gaussian DEFAULT_VARIANCE_LOWER_BOUND
# what a model does when the device reports GPSO_E_PRECISION: reopen the posterior in the next more
# precise arithmetic ON THE DEVICE (never on the CPU, never in the test oracle)
PRECISION_ESCALATION = {"float32": "mixed", "mixed": "float64"}
# ... and when a float32 FACTORISATION loses positive definiteness (GPSO_E_NOTPD inside a hyper-parameter search or a fit at
# the stored hyper-parameters): "mixed" fits in float64, as the reference does; there is no further step
FIT_ESCALATION = {"float32": "mixed"}


def _softplus1(u):
    """softplus(u) = numpy's ``logaddexp(0, u)``, spelled out with its case split on libm's log1p / exp (what numpy's
    scalar loop calls) so that this module and ``gpso_fit_eval_u`` (csrc/api.hip: gpso_softplus) give the same bits."""
    if u == 0.0:
        return 0.693147180559945309417232121458176568
    if u < 0.0:
        return 0.0 + math.log1p(math.exp(u))
    if u > 0.0:
        return u + math.log1p(math.exp(-u))
    return u


def _softplus(u):
    u = np.asarray(u, dtype=np.float64)
    if u.ndim == 0:
        return np.float64(_softplus1(float(u)))
    return np.array([_softplus1(float(v)) for v in u.ravel()], dtype=np.float64).reshape(u.shape)


def _softplus_inv(x):
    x = np.asarray(x, dtype=np.float64)
    return x + np.log(-np.expm1(-x))


def _sigmoid(u):
    """(1 + tanh(u / 2)) / 2 on libm's tanh (the same bits as csrc/api.hip: gpso_sigmoid)."""
    u = np.asarray(u, dtype=np.float64)
    return np.array([0.5 * (1.0 + math.tanh(0.5 * float(v))) for v in u.ravel()], dtype=np.float64).reshape(u.shape)


class _Result(np.ndarray):
    """ndarray that also answers ``.numpy()`` like the tf.Tensor the reference's callers index."""

    def numpy(self):
        return np.asarray(self)


def _as_result(a):
    return np.asarray(a, dtype=np.float64).reshape(-1, 1).view(_Result)


class HipGPR:
    def __init__(self, data, kernel, mean_function=None, noise_variance=1.0e-3, dtype="float64",
                 device=0, engine=None, engine_options=None, escalate=True, devices=None):
        """``dtype``: "float64" | "mixed" | "float32" (see ``HipGPEngine``).  ``engine_options``: extra
        keyword arguments of ``HipGPEngine`` (predict_math, generation, tolerances).  ``escalate``:
        when the device reports that float predictions fail their self-test on the current posterior
        (GPSO_E_PRECISION), reopen it as "mixed", then "float64", with a logged warning, instead of
        raising."""
        if not isinstance(kernel, Kernel):
            raise TypeError("kernel must be a pygpso_amd.kernels.Kernel")
        if mean_function is not None and not isinstance(mean_function, MeanFunction):
            raise TypeError("mean_function must be a pygpso_amd.kernels.MeanFunction or None")
        self.kernel = kernel
        # [gpflow] GPR(mean_function=None) uses Zero(): a fixed mean, nothing to train
        self.mean_function = mean_function if mean_function is not None else Zero()
        self._train_mean = isinstance(self.mean_function, Constant)
        self.likelihood = types.SimpleNamespace(variance=float(noise_variance))
        self._engine_options = dict(engine_options or {})
        self._device = device
        self._devices = list(devices) if devices is not None else None
        self._owns_engine = engine is None
        self.escalate = bool(escalate)
        self.fit_escalations = 0  # hyper-parameter searches restarted on a more precise engine (Scipy.minimize)
        self.fused_transforms = True  # loss evaluations through gpso_fit_eval_u (False: transforms in Python)
        self.engine = engine if engine is not None else self._open_engine(dtype)
        self._data = None
        self._resident = False  # posterior on the device matches (data, hyper-parameters)?
        self.num_loss_evals = 0
        self.data = data

    def _open_engine(self, dtype):
        """One engine on ``device``, or -- ``devices=[...]`` -- a group that shards every predict-type
        call over several GPUs of this process (pygpso_amd/distributed.py)."""
        if self._devices is not None and len(self._devices) > 1:
            from .distributed import HipGPEngineGroup

            return HipGPEngineGroup(dtype=dtype, devices=self._devices, **self._engine_options)
        eng = HipGPEngine(dtype=dtype, device=self._device, **self._engine_options)
        eng.set_timing(False)  # nobody reads last_ms behind the drop-in surrogate: 10 - 45 small evaluations per update
        return eng

    # -- data ---------------------------------------------------------------------------------
    @property
    def data(self):
        return self._data

    @data.setter
    def data(self, value):
        x, y = value
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.ascontiguousarray(y, dtype=np.float64).reshape(-1, 1)
        assert x.ndim == 2 and x.shape[0] == y.shape[0]
        self._data = (x, y)
        self.engine.set_data(x, y[:, 0])
        self._resident = False
        self._device_theta = None  # hyper-parameters of the posterior on the device (None: none / unknown)

    def append_data(self, x_new, y_new):
        """Extend the training data by ``x_new [k, D]``, ``y_new [k]`` or ``[k, 1]`` WITHOUT touching the hyper-parameters
        (``gpso_append``): when the posterior on the device is the one of the current hyper-parameters it is extended in
        place -- O(N^2 k) -- instead of refactorised.  Returns True when the device posterior was extended (or refitted at
        the same hyper-parameters by the library: pad crossing, N <= 128), False for engines without ``append``."""
        x_new = np.ascontiguousarray(np.atleast_2d(x_new), dtype=np.float64)
        y_new = np.ascontiguousarray(y_new, dtype=np.float64).reshape(-1, 1)
        assert x_new.shape[0] == y_new.shape[0] and x_new.shape[1] == self._data[0].shape[1]
        x = np.concatenate([self._data[0], x_new])
        y = np.concatenate([self._data[1], y_new])
        if not hasattr(self.engine, "append"):
            self.data = (x, y)
            return False
        self._ensure_resident()  # the posterior of the data so far at the current hyper-parameters (a fit only if it is not there)
        try:
            nlml, _ = self.engine.append(x_new, y_new[:, 0])
        except np.linalg.LinAlgError:
            # the appended block is not positive definite in this arithmetic (GPSO_E_NOTPD): whatever the device still
            # holds, the model's state must be ONE consistent thing -- the N + k points as data, no resident posterior,
            # nothing known about the device's hyper-parameters -- so that the caller's next step (GPRSurrogate: a
            # re-optimisation, which may reopen a float32 engine as "mixed") starts from a plain ``model.data = (x, y)``
            self.data = (x, y)
            raise
        # only a successful append changes what the model claims
        self._data = (x, y)
        self._last_nlml = nlml
        self._resident = True
        return True

    # -- hyper-parameters ---------------------------------------------------------------------
    @property
    def n_ls(self):
        return int(np.size(self.kernel.lengthscales))

    def _pack(self):
        """Unconstrained vector in tf.Module's sorted order: kernel.lengthscales, kernel.variance,
        likelihood.variance, mean_function.c."""
        parts = [
            np.atleast_1d(_softplus_inv(self.kernel.lengthscales)),
            [float(_softplus_inv(self.kernel.variance))],
            [float(_softplus_inv(self.likelihood.variance - NOISE_FLOOR))],
        ]
        if self._train_mean:
            parts.append([self.mean_function.c])
        return np.concatenate(parts).astype(np.float64)

    def _unpack(self, u):
        u = np.asarray(u, dtype=np.float64)
        k = self.n_ls
        ls = _softplus(u[:k])
        var = float(_softplus(u[k]))
        noise = NOISE_FLOOR + float(_softplus(u[k + 1]))
        c = float(u[k + 2]) if self._train_mean else float(self.mean_function.c)
        return ls, var, noise, c

    def _assign(self, u):
        ls, var, noise, c = self._unpack(u)
        self.kernel.lengthscales = ls.copy() if self.kernel.ard else float(ls[0])
        self.kernel.variance = var
        self.likelihood.variance = noise
        if self._train_mean:
            self.mean_function.c = c
        self._resident = False

    @property
    def trainable_variables(self):
        return self._pack()

    def _theta(self):
        return (self.kernel.name, np.atleast_1d(np.asarray(self.kernel.lengthscales, dtype=np.float64)),
                self.kernel.variance, self.likelihood.variance, float(self.mean_function.c))

    # -- loss ---------------------------------------------------------------------------------
    def _loss_and_grad(self, u):
        """f(u), df/du for L-BFGS-B: one device evaluation (Gram -> Cholesky -> ... -> gradient)."""
        self._device_theta = None
        k = self.n_ls
        if self.fused_transforms and hasattr(self.engine, "fit_eval_u"):
            # transforms + chain rule inside the library: one C-ABI call per evaluation (bit-identical to the branch
            # below: tests/test_gpu_goldens.py::test_loss_evaluation_in_the_optimisers_variables)
            f, gu, th = self.engine.fit_eval_u(self.kernel.name, u, k, self._train_mean, float(self.mean_function.c))
            self._device_theta = self._theta_key(self.kernel.name, th[:k], th[k], th[k + 1], th[k + 2])
            self._last_nlml = f
            self.num_loss_evals += 1
            self._resident = False
            return f, gu
        ls, var, noise, c = self._unpack(u)
        f, g = self.engine.fit_eval(self.kernel.name, ls, var, noise, c, want_grad=True)
        self._device_theta = self._theta_key(self.kernel.name, ls, var, noise, c)
        self._last_nlml = f
        self.num_loss_evals += 1
        self._resident = False  # resident for u, not necessarily for the stored hyper-parameters
        gu = np.empty(k + 2 + (1 if self._train_mean else 0))
        gu[: k + 2] = g[: k + 2] * _sigmoid(np.asarray(u[: k + 2]))
        if self._train_mean:
            gu[k + 2] = g[k + 2]
        return f, gu

    @staticmethod
    def _theta_key(name, ls, var, noise, c):
        return (name, np.asarray(ls, dtype=np.float64).tobytes(), float(var), float(noise), float(c))

    def _ensure_resident(self):
        if not self._resident:
            name, ls, var, noise, c = self._theta()
            # L-BFGS-B's last loss evaluation is, as a rule, at the point it returns: the posterior that
            # evaluation left on the device is then the one asked for (an evaluation with the gradient builds
            # the same factor, L^-1 and alpha, bit for bit) and no further fit is needed
            if self._device_theta != self._theta_key(name, ls, var, noise, c):
                self._last_nlml, _ = self._fitting(lambda: self.engine.fit_eval(name, ls, var, noise, c, want_grad=False))
                self._device_theta = self._theta_key(name, ls, var, noise, c)
            self._resident = True

    def training_loss(self):
        """Negative log marginal likelihood at the current hyper-parameters."""
        self._resident = False
        self._ensure_resident()
        return self._last_nlml

    def log_marginal_likelihood(self):
        return -self.training_loss()

    # -- predict ------------------------------------------------------------------------------
    def _escalate(self, err, fit=False):
        """GPSO_E_PRECISION (predict) or -- ``fit=True`` -- GPSO_E_NOTPD out of a FLOAT factorisation: move data +
        hyper-parameters to an engine of the next more precise arithmetic (still on the device).  Returns False when
        there is nowhere left to go: a fit error has one step only, "float32" -> "mixed" (whose fit is the float64 one,
        the reference's arithmetic: gpso/gp_surrogate.py:490-503 -- a matrix that is not positive definite THERE is the
        caller's to see, as in the reference)."""
        cur = getattr(self.engine, "dtype_name", None)
        nxt = FIT_ESCALATION.get(cur) if fit else PRECISION_ESCALATION.get(cur)
        if nxt is None or not self.escalate or not self._owns_engine:
            return False
        logging.warning(f"{err}; reopening the GP posterior as a {nxt!r} engine on device {self._device}")
        old = self.engine
        self.engine = self._open_engine(nxt)
        old.close()
        x, y = self._data
        self.engine.set_data(x, y[:, 0])
        self._resident = False
        self._device_theta = None
        return True

    def _fitting(self, call):
        """Run a fit-type engine call; a float32 engine whose factorisation is not positive definite is replaced by a
        "mixed" one (float64 fit) and the call repeated -- ``escalate=False`` keeps the raise."""
        while True:
            try:
                return call()
            except np.linalg.LinAlgError as err:
                if not self._escalate(err, fit=True):
                    raise

    def _predicting(self, call):
        """Run a predict-type engine call; a float engine that reports GPSO_E_PRECISION is replaced by
        a more precise one and the call repeated."""
        while True:
            try:
                self._ensure_resident()
                return call()
            except L.GpsoPrecisionError as err:
                if not self._escalate(err):
                    raise

    def predict_y(self, Xnew):
        """(mean [M,1], var [M,1]); the variance includes the likelihood (noise) variance."""
        Xnew = np.asarray(Xnew)
        mean, var = self._predicting(lambda: self.engine.predict(Xnew))
        return _as_result(mean), _as_result(var)

    def predict_f(self, Xnew):
        mean, var = self.predict_y(Xnew)
        return mean, _as_result(np.asarray(var) - self.likelihood.variance)

    def best_ucb(self, Xnew, varsigma, seg_off=None):
        return self._predicting(lambda: self.engine.best_ucb(Xnew, varsigma, seg_off))

    def best_ucb_grow(self, bounds, depth, varsigma):
        return self._predicting(lambda: self.engine.best_ucb_grow(bounds, depth, varsigma))

    # -- reporting ----------------------------------------------------------------------------
    def parameter_dict(self):
        return {
            ".kernel.lengthscales": np.asarray(self.kernel.lengthscales, dtype=np.float64),
            ".kernel.variance": np.float64(self.kernel.variance),
            ".likelihood.variance": np.float64(self.likelihood.variance),
            ".mean_function.c": np.float64(self.mean_function.c),
        }

    def summary(self):
        """Plain-text parameter table in the column order GPflow's print_summary uses."""
        rows = [
            ("GPR.mean_function.c", "", self.mean_function.c),
            ("GPR.kernel.variance", "Softplus", self.kernel.variance),
            ("GPR.kernel.lengthscales", "Softplus", self.kernel.lengthscales),
            ("GPR.likelihood.variance", "Softplus + Shift", self.likelihood.variance),
        ]
        lines = [f"{'name':<24} {'transform':<17} {'value'}"]
        for name, tr, val in rows:
            v = np.array2string(np.asarray(val), precision=6) if np.ndim(val) else f"{val:.6g}"
            lines.append(f"{name:<24} {tr:<17} {v}")
        return "\n".join(lines)

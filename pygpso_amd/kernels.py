"""
Kernel / mean-function / optimiser SPEC objects.

The reference is configured with GPflow objects (``gpflow.kernels.Matern52(lengthscales=..,
variance=..)``, ``gpflow.mean_functions.Constant(c)``, ``gpflow.optimizers.Scipy()`` --
gpso/gp_surrogate.py:393-434).  These are the drop-in stand-ins: they only carry names and
initial hyper-parameter values; the arithmetic is in the HIP kernels.
"""
from __future__ import annotations

import numpy as np
import scipy.optimize


class Kernel:
    """Stationary kernel spec: ``lengthscales`` scalar (isotropic) or [D] (ARD), ``variance``."""

    name = None

    def __init__(self, variance=1.0, lengthscales=1.0):
        self.variance = float(variance)
        ls = np.asarray(lengthscales, dtype=np.float64)
        self.lengthscales = ls.copy() if ls.ndim else float(ls)

    @property
    def ard(self):
        return np.ndim(self.lengthscales) > 0

    def __repr__(self):
        return f"{self.name}(variance={self.variance}, lengthscales={self.lengthscales})"


class Matern52(Kernel):
    name = "Matern52"


class Matern32(Kernel):
    name = "Matern32"


class Matern12(Kernel):
    name = "Matern12"


class SquaredExponential(Kernel):
    name = "SquaredExponential"


RBF = SquaredExponential
Exponential = Matern12

KERNEL_CLASSES = {c.name: c for c in (Matern52, Matern32, Matern12, SquaredExponential)}


class MeanFunction:
    pass


class Constant(MeanFunction):
    """m(x) = c"""

    name = "Constant"

    def __init__(self, c=0.0):
        self.c = float(np.asarray(c).reshape(-1)[0]) if np.ndim(c) else float(c)

    def __repr__(self):
        return f"Constant(c={self.c})"


class Zero(MeanFunction):
    name = "Zero"
    c = 0.0


class Scipy:
    """L-BFGS-B through SciPy with SciPy's defaults -- what
    ``gpflow.optimizers.Scipy().minimize(model.training_loss, model.trainable_variables)`` does
    (gpso/gp_surrogate.py:500-503).  ``closure`` must be the bound ``training_loss`` of a model that
    offers ``_loss_and_grad(u)`` / ``_pack()`` / ``_assign(u)`` (pygpso_amd.model.HipGPR).

    With the defaults the L-BFGS-B routine itself (``scipy.optimize._lbfgsb.setulb``, the code
    ``scipy.optimize.minimize`` drives) is called in the same reverse-communication loop
    ``scipy.optimize._lbfgsb_py._minimize_lbfgsb`` runs, minus the per-evaluation wrappers of
    ``minimize`` (``ScalarFunction``, ``OptimizeResult`` per iteration): same routine, same inputs, same
    iterates bit for bit -- at N <= 100 those wrappers cost as much as the device evaluation.  Any option,
    another method, or a SciPy whose private routine has another signature goes through
    ``scipy.optimize.minimize``."""

    _SETULB_DOC = "setulb(m,x,l,u,nbd,f,g,factr,pgtol,wa,iwa,task,lsave,isave,dsave,maxls,ln_task)"

    def minimize(self, closure, variables=None, method="L-BFGS-B", **scipy_kwargs):
        model = getattr(closure, "__self__", None)
        if model is None or not hasattr(model, "_loss_and_grad"):
            raise TypeError("Scipy.minimize expects the bound training_loss of a HipGPR model")
        x0 = model._pack()  # (the model's hyper-parameters are assigned only at the end: a restart begins here again)
        while True:
            try:
                res = None
                if method == "L-BFGS-B" and not scipy_kwargs:
                    res = self._lbfgsb_direct(model._loss_and_grad, x0)
                if res is None:
                    res = scipy.optimize.minimize(model._loss_and_grad, x0, jac=True, method=method, **scipy_kwargs)
                break
            except np.linalg.LinAlgError as err:
                # GPSO_E_NOTPD inside the search.  The reference's search runs in float64 (gpflow.default_float,
                # gpso/gp_surrogate.py:490-503) and loses positive definiteness only on a genuinely singular matrix; a
                # float32 factorisation loses it where the line search steps into small noise at large N.  The model then
                # reopens its engine as "mixed" (float64 fit, on the device) and THIS update's search starts over from its
                # theta_0 -- one history in one arithmetic.  Nowhere to go (float64 / mixed engine, escalate=False, an engine
                # the model does not own): the error is the caller's, as in the reference.
                escalate = getattr(model, "_escalate", None)
                if escalate is None or not escalate(err, fit=True):
                    raise
                model.fit_escalations = getattr(model, "fit_escalations", 0) + 1
        model._assign(res.x)
        return res

    @classmethod
    def _lbfgsb_direct(cls, fun_and_grad, x0):
        """The loop of scipy.optimize._lbfgsb_py._minimize_lbfgsb (SciPy 1.15) for an unbounded problem
        with an exact gradient and default options; None when the private routine is not the expected one."""
        try:
            from scipy.optimize import _lbfgsb
        except ImportError:
            return None
        if (getattr(_lbfgsb.setulb, "__doc__", None) or "").strip() != cls._SETULB_DOC:
            return None
        m, maxls, maxfun, maxiter = 10, 20, 15000, 15000
        factr = 2.2204460492503131e-09 / np.finfo(float).eps
        pgtol = 1e-5
        x = np.array(np.asarray(x0).ravel(), dtype=np.float64)
        n = x.shape[0]
        nbd = np.zeros(n, np.int32)
        low_bnd = np.zeros(n, np.float64)
        upper_bnd = np.zeros(n, np.float64)
        f = np.array(0.0, dtype=np.int32)
        g = np.zeros((n,), dtype=np.int32)
        wa = np.zeros(2 * m * n + 5 * n + 11 * m * m + 8 * m, np.float64)
        iwa = np.zeros(3 * n, dtype=np.int32)
        task = np.zeros(2, dtype=np.int32)
        ln_task = np.zeros(2, dtype=np.int32)
        lsave = np.zeros(4, dtype=np.int32)
        isave = np.zeros(44, dtype=np.int32)
        dsave = np.zeros(29, dtype=np.float64)
        nfev = nit = 0
        while True:
            g = g.astype(np.float64)
            _lbfgsb.setulb(m, x, low_bnd, upper_bnd, nbd, f, g, factr, pgtol, wa, iwa, task, lsave, isave,
                           dsave, maxls, ln_task)
            if task[0] == 3:  # the routine wants f and g at the current x
                f, g = fun_and_grad(np.copy(x))
                f = float(f)
                g = np.array(g, dtype=np.float64).ravel()
                nfev += 1
            elif task[0] == 1:  # new iteration
                nit += 1
                if nit >= maxiter:
                    task[0], task[1] = 5, 504
                elif nfev > maxfun:
                    task[0], task[1] = 5, 502
            else:
                break
        status = 0 if task[0] == 4 else (1 if (nfev > maxfun or nit >= maxiter) else 2)
        return scipy.optimize.OptimizeResult(fun=f, jac=g, nfev=nfev, njev=nfev, nit=nit, status=status,
                                             x=x, success=(status == 0),
                                             message=f"L-BFGS-B task {int(task[0])}/{int(task[1])}")

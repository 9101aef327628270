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
    """L-BFGS-B through ``scipy.optimize.minimize`` with SciPy defaults -- what
    ``gpflow.optimizers.Scipy().minimize(model.training_loss, model.trainable_variables)`` does
    (gpso/gp_surrogate.py:500-503).  ``closure`` must be the bound ``training_loss`` of a model that
    offers ``_loss_and_grad(u)`` / ``_pack()`` / ``_assign(u)`` (pygpso_amd.model.HipGPR)."""

    def minimize(self, closure, variables=None, method="L-BFGS-B", **scipy_kwargs):
        model = getattr(closure, "__self__", None)
        if model is None or not hasattr(model, "_loss_and_grad"):
            raise TypeError("Scipy.minimize expects the bound training_loss of a HipGPR model")
        res = scipy.optimize.minimize(model._loss_and_grad, model._pack(), jac=True, method=method,
                                      **scipy_kwargs)
        model._assign(res.x)
        return res

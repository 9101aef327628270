"""
pygpso_amd -- MI355X-native GP-surrogate + ternary-tree acquisition engine: the hot path of
jajcayn/pygpso (GP fit + per-leaf UCB predict) as hand-written HIP kernels behind a ctypes C-ABI.

Public names mirror the reference package (``gpso/__init__.py:10-12``).
"""
__version__ = "0.1.0"

from .engine import HipGPEngine  # noqa: F401
from .gp_surrogate import GPListOfPoints, GPPoint, GPRSurrogate, GPSurrogate  # noqa: F401
from .grids import conditional_surrogate_grids  # noqa: F401
from .optimisation import GPSOCallback, GPSOptimiser  # noqa: F401
from .param_space import LeafNode, ParameterSpace  # noqa: F401
from .utils import PointLabels  # noqa: F401

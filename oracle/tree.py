"""
float64 restatement of the ternary partition geometry and the MinMax normalisation.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``) -- never imported by ``pygpso_amd``.

Follows (paths relative to /root/reference):
* ``LeafNode.ternary_split`` ........ gpso/param_space.py:257-307  (widest dim by first arg-max,
                                      cut points ``lo + i * (width / 3)`` for i = 0..3)
* ``LeafNode.get_center_as_list`` ... gpso/param_space.py:202-217  (``np.mean`` of each [lo, hi])
* ``LeafNode.grow`` ................. gpso/param_space.py:175-200  (level-major list of centres,
                                      levels 0..depth-1, every node expanded into l, c, r in order)
* ``LeafNode.sample_uniformly`` ..... gpso/param_space.py:157-173
* ``ParameterSpace`` scaler ......... gpso/param_space.py:371-372,424-450 with
                                      [sklearn] ``MinMaxScaler(feature_range=(0, 1))`` semantics
                                      

Pure-Python float arithmetic on purpose: Python floats are IEEE doubles, so this reproduces the
reference's rounding exactly (it also computes on Python floats / numpy float64 scalars).
"""
from __future__ import annotations

import numpy as np


def split_bounds(bounds):
    """bounds: list of (lo, hi) python floats -> three child bounds lists (l, c, r)."""
    widths = [b[1] - b[0] for b in bounds]
    k = int(np.argmax(widths))  # first maximum
    delta = widths[k] / 3
    cuts = [bounds[k][0] + i * delta for i in range(4)]
    children = []
    for j in range(3):
        child = list(bounds)
        child[k] = (cuts[j], cuts[j + 1])
        children.append(child)
    return children


def centre(bounds):
    return [float(np.mean(b)) for b in bounds]


def grow(bounds, depth: int) -> np.ndarray:
    """Centres of levels 0..depth-1 of the ternary subtree under ``bounds``; [(3^depth-1)/2, D]."""
    bounds = [(float(lo), float(hi)) for lo, hi in bounds]
    level = [bounds]
    rows = []
    for _ in range(depth):
        rows.extend(centre(b) for b in level)
        nxt = []
        for b in level:
            nxt.extend(split_bounds(b))
        level = nxt
    return np.array(rows, dtype=np.float64).reshape(-1, len(bounds))


def grow_count(depth: int) -> int:
    return (3**depth - 1) // 2


def sample_uniformly(bounds, n_points: int, seed=None) -> np.ndarray:
    np.random.seed(seed)
    return np.random.uniform(
        low=[b[0] for b in bounds], high=[b[1] for b in bounds], size=(n_points, len(bounds))
    )


class MinMax01:
    """[sklearn] MinMaxScaler(feature_range=(0,1)) fitted on the two rows (lows, highs)."""

    def __init__(self, parameter_bounds):
        pb = np.asarray(parameter_bounds, dtype=np.float64)
        data_min, data_max = pb[:, 0], pb[:, 1]
        self.scale_ = (1.0 - 0.0) / (data_max - data_min)
        self.min_ = 0.0 - data_min * self.scale_

    def transform(self, x):
        x = np.array(x, dtype=np.float64)
        x *= self.scale_
        x += self.min_
        return x

    def inverse_transform(self, x):
        x = np.array(x, dtype=np.float64)
        x -= self.min_
        x /= self.scale_
        return x

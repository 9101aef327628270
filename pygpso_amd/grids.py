"""
Conditional-surrogate grids: the second consumer of ``predict_y`` in the reference
(``gpso/plotting.py:257-381`` evaluates the surrogate on a granularity x granularity slice through
the best point for every pair of parameters, one ``predict_y`` call per pair).  Here all
D (D - 1) / 2 slices go to the device as ONE batch; the numbers are what the reference plots, the
plotting itself is out of scope.
"""
from __future__ import annotations

import numpy as np


def conditional_surrogate_grids(gp_surr, granularity=100, through=None):
    """{(i, j): (mean[g, g], var[g, g])} for every parameter pair i < j.

    ``through``: normalised coordinates the slices pass through (default: the point with the
    highest evaluated score).  Grid layout as in the reference: ``x`` (parameter i) runs along the
    second axis, ``y`` (parameter j) along the first (``np.meshgrid`` order); ``var`` includes the
    noise variance.
    """
    if through is None:
        through = gp_surr.highest_score.normed_coord
    through = np.asarray(through, dtype=np.float64).reshape(-1)
    d = through.shape[0]
    g = int(granularity)
    axis = np.linspace(0, 1, g)
    xg, yg = np.meshgrid(axis, axis)
    xf, yf = xg.flatten(), yg.flatten()
    pairs = [(i, j) for i in range(d) for j in range(i + 1, d)]
    if not pairs:
        return {}
    batch = np.tile(through, (len(pairs) * g * g, 1))
    for p, (i, j) in enumerate(pairs):
        rows = slice(p * g * g, (p + 1) * g * g)
        batch[rows, i] = xf
        batch[rows, j] = yf
    mean, var = gp_surr.gpflow_model.predict_y(batch)
    mean, var = np.asarray(mean)[:, 0], np.asarray(var)[:, 0]
    return {pair: (mean[p * g * g:(p + 1) * g * g].reshape(g, g), var[p * g * g:(p + 1) * g * g].reshape(g, g))
            for p, pair in enumerate(pairs)}

"""
Parameter space and its ternary partition tree -- producer of the leaf batches the GP scores.

Mirrors the reference's interface (gpso/param_space.py): ``LeafNode`` (:20-307: ``ternary_split``,
``grow``, ``sample_uniformly``, centres) and ``ParameterSpace`` (:310-467: normalisation,
``max_depth``, ``get_best_score_leaf``, save/load).  Built differently:

* no anytree / sklearn: a node keeps ``parent`` / ``children`` / ``depth`` / its child-index path,
  and the root keeps one node list per depth, so ``get_best_score_leaf`` scans only that level
  (pre-order among equals == lexicographic order of paths) instead of walking the whole tree;
* ``grow`` expands a whole level at a time with numpy (same IEEE-754 operations in the same order
  as the reference's per-node Python arithmetic, hence bit-identical rows) and never builds node
  objects; the device twin is ``gpso_grow`` / ``gpso_best_ucb_grow`` (csrc/grow.hip).
"""
from __future__ import annotations

import pickle

import numpy as np

from .utils import PKL_EXT, PointLabels

NORM_PARAMS_BOUNDS = (0, 1)


class MinMaxScaler01:
    """Affine map of the parameter box onto [0, 1]^D with sklearn ``MinMaxScaler`` arithmetic
    (``x * scale_ + min_`` / ``(x - min_) / scale_``)."""

    def __init__(self, parameter_bounds):
        pb = np.asarray(parameter_bounds, dtype=np.float64)
        self.data_min_ = pb[:, 0].copy()
        self.data_max_ = pb[:, 1].copy()
        self.scale_ = (NORM_PARAMS_BOUNDS[1] - NORM_PARAMS_BOUNDS[0]) / (self.data_max_ - self.data_min_)
        self.min_ = NORM_PARAMS_BOUNDS[0] - self.data_min_ * self.scale_

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


def _split_level(lo, hi):
    """Ternary split of every box of a level: lo, hi [n, D] -> lo, hi [3n, D] in (l, c, r) order.
    widths -> first arg-max -> delta = w / 3 -> cuts lo_k + i * delta (gpso/param_space.py:272-277)."""
    n = lo.shape[0]
    widths = hi - lo
    k = np.argmax(widths, axis=1)
    rows = np.arange(n)
    delta = widths[rows, k] / 3
    base = lo[rows, k]
    cuts = [base + i * delta for i in range(4)]
    lo3 = np.repeat(lo, 3, axis=0)
    hi3 = np.repeat(hi, 3, axis=0)
    for j in range(3):
        lo3[j::3][rows, k] = cuts[j]
        hi3[j::3][rows, k] = cuts[j + 1]
    return lo3, hi3


class LeafNode:
    """A hyper-rectangular cell of the partition, in normalised coordinates."""

    def __init__(self, norm_bounds, scaler, parameter_names, score=0.0, sampled=False,
                 label=PointLabels.not_assigned, name="", parent=None, children=None):
        assert isinstance(norm_bounds, (list, tuple)) and norm_bounds is not None
        for b in norm_bounds:
            assert isinstance(b, (list, tuple)) and len(b) == 2 and b[1] > b[0]
        self.scaler = scaler
        assert len(norm_bounds) == self.ndim
        assert len(parameter_names) == self.ndim
        assert all(isinstance(p, str) for p in parameter_names)
        self.norm_bounds = norm_bounds
        self.parameter_names = parameter_names
        self.name = name
        self.score = score
        self.sampled = sampled
        self.label = label
        self.children = []
        self.parent = None
        self.depth = 0
        self.path = ()
        self.point_index = None  # index of this cell's centre in the surrogate's point store
        if parent is not None:
            parent._attach(self)
        for ch in children or []:
            self._attach(ch)

    # -- tree plumbing ------------------------------------------------------------------------
    def _attach(self, child):
        child.parent = self
        child.depth = self.depth + 1
        child.path = self.path + (len(self.children),)
        self.children.append(child)
        self.root._register(child)

    @property
    def root(self):
        node = self
        while node.parent is not None:
            node = node.parent
        return node

    def _register(self, node):  # only meaningful on a ParameterSpace root
        pass

    def iter_preorder(self):
        stack = [self]
        while stack:
            node = stack.pop()
            yield node
            stack.extend(reversed(node.children))

    def __getitem__(self, pos):
        return self.children[pos]

    def __str__(self):
        return (f"Leaf node `{self.name}`: score {self.score}; center at "
                f"{self.get_center_as_dict(normed=True)}; depth {self.depth}")

    __repr__ = __str__

    # -- geometry -----------------------------------------------------------------------------
    @property
    def ndim(self):
        return self.scaler.data_max_.shape[0]

    def bounds_array(self):
        """[D, 2] float64 (lo, hi) -- the form the device generator takes."""
        return np.array([[b[0], b[1]] for b in self.norm_bounds], dtype=np.float64)

    def get_center_as_list(self, normed=False):
        # (lo + hi) / 2 is what np.mean of the pair computes, bit for bit, without 1 000s of ufunc set-ups
        centers = [(float(b[0]) + float(b[1])) / 2.0 for b in self.norm_bounds]
        if not normed:
            centers = np.around(self.scaler.inverse_transform(np.array([centers])), decimals=5)[0].tolist()
        return centers

    def get_center_as_dict(self, normed=False):
        return dict(zip(self.parameter_names, self.get_center_as_list(normed=normed)))

    def sample_uniformly(self, n_points, seed=None):
        np.random.seed(seed)
        return np.random.uniform(low=[b[0] for b in self.norm_bounds],
                                 high=[b[1] for b in self.norm_bounds], size=(n_points, self.ndim))

    def grow(self, depth):
        """Centres of levels 0..depth-1 of the ternary sub-tree under this cell, level-major,
        [(3^depth - 1) / 2, D]; the tree itself is left untouched."""
        b = self.bounds_array()
        lo, hi = b[None, :, 0].copy(), b[None, :, 1].copy()
        out = []
        for _ in range(depth):
            out.append((lo + hi) / 2)
            lo, hi = _split_level(lo, hi)
        if not out:
            return np.empty((0, self.ndim))
        return np.vstack(out)

    def ternary_split(self):
        """Split along the widest dimension (first arg-max) into thirds; attaches and returns the
        children ``l``, ``c``, ``r``.  The centre child shares this cell's centre."""
        widths = [b[1] - b[0] for b in self.norm_bounds]
        k = int(np.argmax(widths))
        delta = widths[k] / 3
        cuts = [self.norm_bounds[k][0] + i * delta for i in range(4)]
        kids = []
        for j, tag in enumerate("lcr"):
            bounds = [b if idx != k else (cuts[j], cuts[j + 1]) for idx, b in enumerate(self.norm_bounds)]
            kids.append(LeafNode(norm_bounds=bounds, scaler=self.scaler,
                                 parameter_names=self.parameter_names, name=self.name + "->" + tag,
                                 parent=self))
        np.testing.assert_allclose(kids[1].get_center_as_list(normed=True),
                                   self.get_center_as_list(normed=True))
        return kids

    # -- (de)serialisation ----------------------------------------------------------------------
    def _to_dict(self):
        return {
            "norm_bounds": [tuple(b) for b in self.norm_bounds],
            "name": self.name, "score": self.score, "sampled": self.sampled,
            "label": self.label.name, "point_index": self.point_index,
            "children": [c._to_dict() for c in self.children],
        }


class ParameterSpace(LeafNode):
    """Root of the partition tree: the whole (normalised) parameter box."""

    def __init__(self, parameter_bounds, parameter_names):
        assert parameter_bounds is not None and isinstance(parameter_bounds, (list, tuple))
        for b in parameter_bounds:
            assert isinstance(b, (list, tuple)) and len(b) == 2 and b[1] > b[0]
        scaler = MinMaxScaler01(parameter_bounds)
        parameter_names = parameter_names or ["" for _ in range(len(parameter_bounds))]
        assert len(parameter_names) == len(parameter_bounds)
        self._levels = [[self]]
        super().__init__(norm_bounds=[NORM_PARAMS_BOUNDS for _ in parameter_bounds], scaler=scaler,
                         parameter_names=parameter_names, name="full_domain")

    def _register(self, node):
        while len(self._levels) <= node.depth:
            self._levels.append([])
        self._levels[node.depth].append(node)

    @property
    def max_depth(self):
        return len(self._levels) - 1

    def get_best_score_leaf(self, depth, only_not_sampled=True):
        """Highest-scored cell of a level (optionally only cells not yet split); ties go to the
        first in pre-order, like the reference's stable sort (gpso/param_space.py:399-422)."""
        if depth >= len(self._levels):
            return None
        best = None
        for node in self._levels[depth]:
            if node.sampled and only_not_sampled:
                continue
            if best is None or node.score > best.score or (node.score == best.score and node.path < best.path):
                best = node
        return best

    def normalise_coords(self, orig_coords):
        assert orig_coords.ndim == 2 and orig_coords.shape[1] == self.ndim
        return self.scaler.transform(orig_coords)

    def denormalise_coords(self, normed_coords):
        assert normed_coords.ndim == 2 and normed_coords.shape[1] == self.ndim
        return self.scaler.inverse_transform(normed_coords)

    # -- persistence ------------------------------------------------------------------------------
    def save(self, filename):
        if not filename.endswith(PKL_EXT):
            filename += PKL_EXT
        payload = {
            "parameter_bounds": np.stack([self.scaler.data_min_, self.scaler.data_max_], axis=1).tolist(),
            "parameter_names": list(self.parameter_names),
            "tree": self._to_dict(),
        }
        with open(filename, "wb") as fh:
            pickle.dump(payload, fh, protocol=pickle.HIGHEST_PROTOCOL)

    @classmethod
    def from_file(cls, filename):
        if not filename.endswith(PKL_EXT):
            filename += PKL_EXT
        with open(filename, "rb") as fh:
            payload = pickle.load(fh)
        space = cls([list(b) for b in payload["parameter_bounds"]], payload["parameter_names"])

        def restore(node, rec):
            node.score, node.sampled = rec["score"], rec["sampled"]
            node.label = PointLabels[rec["label"]]
            node.point_index = rec.get("point_index")
            for ch in rec["children"]:
                kid = LeafNode(norm_bounds=[tuple(b) for b in ch["norm_bounds"]], scaler=space.scaler,
                               parameter_names=space.parameter_names, name=ch["name"], parent=node)
                restore(kid, ch)

        restore(space, payload["tree"])
        return space

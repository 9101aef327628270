"""
GPSO loop: the caller of the hot path (update -> explore -> select).

Mirrors ``GPSOptimiser`` of the reference (gpso/optimisation.py:54-718) -- same constructor
arguments, ``run`` / ``resume_run`` / ``save_state`` / ``resume_from_saved``, and the same three
steps with the same quirks:

* ``_gp_update`` ...... optimisation.py:314-340   retrain + re-score every gp_based cell
* ``_tree_explore`` ... optimisation.py:342-403   split best cell per flagged level, score children
* ``_tree_select`` .... optimisation.py:405-462   evaluate the objective at promising cells

MI355X-first differences (results unchanged): both new children of a level are scored by ONE
device call that also generates their ternary sub-trees on the GPU
(``GPRSurrogate.gp_eval_best_ucb_grow`` -> ``gpso_best_ucb_grow``); cells remember the index of
their centre in the point store instead of re-searching it on every update; objective workers are
spawned (never forked) because a process that has initialised HIP must not fork.
"""
from __future__ import annotations

import json
import logging
import os
from enum import Enum, unique
from functools import partial

import numpy as np

from .gp_surrogate import GPPoint, GPRSurrogate, GPSurrogate
from .param_space import NORM_PARAMS_BOUNDS, ParameterSpace
from .utils import JSON_EXT, PKL_EXT, PointLabels


@unique
class CallbackTypes(Enum):
    post_initialise = "post_initialise"
    pre_iteration = "pre_iteration"
    post_iteration = "post_iteration"
    post_update = "post_update"
    pre_finalise = "pre_finalise"


class GPSOCallback:
    """Base class for user hooks: set ``callback_type`` and implement ``run(optimiser)``."""

    callback_type = None

    def run(self, optimiser):
        assert isinstance(optimiser, GPSOptimiser)


class GPSOptimiser:
    SAVE_ATTRS = ["iterations", "budget", "eval_repeats", "last_explored_levels", "last_update_idx",
                  "method", "max_depth", "stop_cond", "update_cycle", "n_eval_counter", "n_workers",
                  "expl_seed"]
    PARAM_SPACE_FILE = f"parameter_space{PKL_EXT}"
    OPT_ATTRS_FILE = f"opt_attributes{JSON_EXT}"

    def __init__(self, parameter_space, gp_surrogate=None, exploration_method="tree",
                 exploration_depth=5, budget=100, stopping_condition="evaluations", update_cycle=1,
                 n_workers=1, callbacks=None, saver=None):
        assert isinstance(parameter_space, ParameterSpace)
        self.param_space = parameter_space
        self.method = exploration_method
        if self.method == "tree":
            self.max_depth = exploration_depth
        elif self.method == "sample":
            self.max_depth = exploration_depth * self.param_space.ndim ** 2
        else:
            raise ValueError(f"Unknown exploration method: {self.method}")
        self.budget = budget
        assert stopping_condition in ["evaluations", "iterations", "depth"]
        self.stop_cond = stopping_condition
        self.update_cycle = update_cycle
        self.n_eval_counter = 0
        self.iterations = 0
        self.n_workers = n_workers
        callbacks = callbacks or []
        assert all(isinstance(cb, GPSOCallback) for cb in callbacks)
        self.callbacks = callbacks
        self.gp_surr = gp_surrogate or GPRSurrogate.default()
        assert isinstance(self.gp_surr, GPSurrogate)
        self.saver = saver
        if saver is not None:
            assert callable(getattr(saver, "save_runs", None))
        # batch both children of a level into one on-device grow+score call when the surrogate can
        self.device_grow = hasattr(self.gp_surr, "gp_eval_best_ucb_grow")

    # -- helpers ---------------------------------------------------------------------------------
    def _run_callbacks(self, callback_type):
        assert callback_type in CallbackTypes
        for cb in self.callbacks:
            if cb.callback_type == callback_type:
                cb.run(self)

    @staticmethod
    def _centre(node):
        return np.array(node.get_center_as_list(normed=True))

    def _point_of(self, node):
        pts = self.gp_surr.points
        if node.point_index is None and hasattr(pts, "find_index_by_coords"):
            node.point_index = pts.find_index_by_coords(self._centre(node))
        if node.point_index is not None:
            return pts[node.point_index]
        return pts.find_by_coords(self._centre(node))

    # -- initial design ----------------------------------------------------------------------------
    def _initialise(self, init_samples):
        d = self.param_space.ndim
        mid, rad = np.mean(NORM_PARAMS_BOUNDS), np.sum(NORM_PARAMS_BOUNDS) * 0.25
        if init_samples is None:
            logging.info("Sampling 2 vertices per dimension within L1 ball of 0.25 of the domain size "
                         f"radius in normalised coordinates using {self.n_workers} worker(s)...")
            normed = np.vstack([mid - rad * np.eye(d), mid + rad * np.eye(d)])
            orig_coords = self.param_space.denormalise_coords(normed)
        elif isinstance(init_samples, np.ndarray):
            assert init_samples.ndim == 2 and init_samples.shape[1] == d
            if init_samples.shape[0] <= 2:
                logging.warning(f"Only {init_samples.shape[0]} points selected for sampling, you "
                                "might want to add more...")
            elif init_samples.shape[0] > 2 * d:
                logging.warning("Too many initial points obtained, you will run out of budget of "
                                "objective function evaluations!")
            logging.info(f"Got {init_samples.shape[0]} points for initial sampling. Note that these "
                         "are interpreted in the original parameter space coordinates!")
            orig_coords = init_samples.copy()
        else:
            raise TypeError("init_samples must be None or a numpy array of original coordinates")
        centre_orig = self.param_space.denormalise_coords(np.array([[mid] * d]))
        all_coords = np.vstack([orig_coords, centre_orig])
        all_scores = self.evaluate_objective_function(all_coords)
        self.param_space.score = float(all_scores[-1])
        self.param_space.label = PointLabels.evaluated
        self.gp_surr.append(self.param_space.normalise_coords(all_coords), all_scores)
        logging.debug(f"Initialised with {all_coords.shape[0]} points")

    # -- update --------------------------------------------------------------------------------
    def _gp_update(self, update_idx):
        if (self.gp_surr.num_evaluated - update_idx) >= self.update_cycle:
            logging.info("Update step: retraining GP model and updating scores...")
            self.gp_surr.gp_update()
            for node in self.param_space.iter_preorder():
                point = self._point_of(node)
                assert point is not None
                if point.label == PointLabels.gp_based:
                    node.score = point.score_ucb
            self._run_callbacks(CallbackTypes.post_update)
        return self.gp_surr.num_evaluated

    # -- explore -------------------------------------------------------------------------------
    def _score_children(self, fresh, kwargs):
        """best (mean, var, ucb) over the exploration samples of each fresh child."""
        if not fresh:
            return []
        if self.method == "tree":
            if self.device_grow:
                bounds = np.stack([ch.bounds_array() for ch in fresh])
                return self.gp_surr.gp_eval_best_ucb_grow(bounds, self.max_depth)
            return [self.gp_surr.gp_eval_best_ucb(ch.grow(depth=self.max_depth)) for ch in fresh]
        out = []
        for ch in fresh:  # "sample": only the first child of a pass sees the seed (reference quirk)
            coords = ch.sample_uniformly(n_points=self.max_depth, seed=kwargs.pop("seed", None))
            out.append(self.gp_surr.gp_eval_best_ucb(coords))
        return out

    def _tree_explore(self, levels_to_explore, **kwargs):
        logging.info("Exploration step: sampling children in the ternary tree...")
        n_levels = self.param_space.max_depth + 1
        assert len(levels_to_explore) == n_levels
        points = self.gp_surr.points
        for level in range(n_levels):
            if not levels_to_explore[level]:
                continue
            logging.debug(f"Exploring {level} level...")
            parent = self.param_space.get_best_score_leaf(depth=level)
            children = parent.ternary_split()
            centres = [self._centre(ch) for ch in children]
            known = [points.find_index_by_coords(c) for c in centres]
            fresh = [ch for ch, idx in zip(children, known) if idx is None]
            scores = iter(self._score_children(fresh, kwargs))
            for ch, c, idx in zip(children, centres, known):
                if idx is None:
                    mu, var, ucb = next(scores)
                    ch.score = ucb
                    ch.label = PointLabels.gp_based
                    # stored at the child's CENTRE, with the values of its best sub-leaf
                    ch.point_index = points.append(GPPoint(c, mu, var, ucb, PointLabels.gp_based))
                else:
                    ch.score = parent.score
                    ch.label = parent.label
                    ch.point_index = idx
                logging.debug(f"{ch.name} best score: {ch.score}")
            parent.sampled = True

    # -- select --------------------------------------------------------------------------------
    def _tree_select(self):
        logging.info("Selecting step: evaluating best leaves...")
        max_score = -np.inf
        n_levels = self.param_space.max_depth + 1
        levels_to_explore = [False] * n_levels
        for level in range(n_levels):
            leaf = self.param_space.get_best_score_leaf(depth=level, only_not_sampled=True)
            if leaf is None or not leaf.score > max_score:
                continue
            levels_to_explore[level] = True
            max_score = float(leaf.score)  # the pre-evaluation score drives the comparison
            point = self._point_of(leaf)
            if point.label == PointLabels.gp_based:
                new_score = float(self.evaluate_objective_function(
                    self.param_space.denormalise_coords(point.normed_coord[np.newaxis, :]))[0])
                self.gp_surr.points.append(
                    GPPoint(point.normed_coord, new_score, 0.0, 0.0, PointLabels.evaluated))
                leaf.score = new_score
                leaf.label = PointLabels.evaluated
                logging.debug(f"Leaf {leaf.name} updated to new evaluated score: {leaf.score}")
        logging.debug(f"Level to explore in the next iteration: {levels_to_explore}")
        return levels_to_explore

    # -- objective -----------------------------------------------------------------------------
    def evaluate_objective_function(self, orig_coords):
        assert orig_coords.ndim == 2 and orig_coords.shape[1] == self.param_space.ndim
        repeated = np.vstack(self.eval_repeats * [orig_coords])
        if self.n_workers > 1 and repeated.shape[0] > 1:
            import multiprocessing as mp
            from concurrent.futures import ProcessPoolExecutor

            # spawn, never fork: this process may already hold a HIP context
            with ProcessPoolExecutor(self.n_workers, mp_context=mp.get_context("spawn")) as pool:
                scores = list(pool.map(self.obj_func, repeated))
        else:
            scores = [self.obj_func(c) for c in repeated]
        self.n_eval_counter += orig_coords.shape[0]
        if self.saver is not None:
            results = [s[0] for s in scores]
            scores = [s[1] for s in scores]
            n = orig_coords.shape[0]
            for i, coords in enumerate(orig_coords):
                self.saver.save_runs(results[i::n], scores[i::n],
                                     dict(zip(self.param_space.parameter_names, coords)))
        return self.eval_repeats_function(
            np.array(scores).astype(float).reshape((self.eval_repeats, -1)))

    def _stopping_condition(self):
        if self.stop_cond == "evaluations":
            return self.n_eval_counter < self.budget
        if self.stop_cond == "iterations":
            return self.iterations < self.budget
        return self.param_space.max_depth <= self.budget

    # -- main loops ------------------------------------------------------------------------------
    def _iterate(self, explore_levels, update_idx):
        cond = True
        while cond:
            self._run_callbacks(CallbackTypes.pre_iteration)
            self._tree_explore(levels_to_explore=explore_levels, seed=self.expl_seed)
            explore_levels = self._tree_select()
            update_idx = self._gp_update(update_idx)
            self.iterations += 1
            highest_ucb = self.gp_surr.highest_ucb
            logging.info(
                f"After {self.iterations}th iteration: \n\t number of obj. func. evaluations: "
                f"{self.n_eval_counter} \n\t highest score: {self.gp_surr.highest_score.score_mu} "
                f"\n\t highest UCB: {highest_ucb.score_ucb if highest_ucb else None}")
            self.trace.append((self.n_eval_counter, self.gp_surr.highest_score.score_mu,
                               highest_ucb.score_ucb if highest_ucb else None))
            self._run_callbacks(CallbackTypes.post_iteration)
            cond = self._stopping_condition()
        logging.info(f"Done. Highest evaluated score: {self.gp_surr.highest_score.score_mu}")
        self._run_callbacks(CallbackTypes.pre_finalise)
        self.last_explored_levels = explore_levels
        self.last_update_idx = update_idx
        return self.gp_surr.highest_score

    def run(self, objective_function, init_samples=None, eval_repeats=1, eval_repeats_function=np.mean,
            **kwargs):
        assert callable(objective_function)
        self.obj_func = objective_function
        self.eval_repeats = eval_repeats
        assert callable(eval_repeats_function)
        self.eval_repeats_function = partial(eval_repeats_function, axis=0)
        self.expl_seed = kwargs.pop("seed", None)
        self.trace = []  # (evaluations, highest score, highest UCB) per iteration
        logging.info(f"Starting {self.param_space.ndim}-dimensional optimisation with budget of "
                     f"{self.budget} objective function evaluations...")
        self._initialise(init_samples)
        self._run_callbacks(CallbackTypes.post_initialise)
        update_idx = self._gp_update(0)
        return self._iterate([True], update_idx)

    def resume_run(self, additional_budget):
        assert callable(self.obj_func) and callable(self.eval_repeats_function)
        assert self.iterations > 0
        self.budget += additional_budget
        if not hasattr(self, "trace"):
            self.trace = []
        logging.info(f"Resuming optimisation for with additional budget of {additional_budget}")
        return self._iterate(self.last_explored_levels, self.last_update_idx)

    # -- persistence -----------------------------------------------------------------------------
    def save_state(self, folder):
        os.makedirs(folder, exist_ok=True)
        logging.warning("When saving, all callbacks and saver will be lost!")
        self.param_space.save(os.path.join(folder, self.PARAM_SPACE_FILE))
        self.gp_surr.save(folder)
        attrs = {a: getattr(self, a) for a in self.SAVE_ATTRS}
        with open(os.path.join(folder, self.OPT_ATTRS_FILE), "w") as fh:
            fh.write(json.dumps(attrs))
        logging.info(f"Saved optimiser to {folder}")

    @classmethod
    def resume_from_saved(cls, folder, additional_budget, objective_function, gp_surrogate=GPRSurrogate,
                          eval_repeats_function=np.mean, callbacks=None, saver=None):
        space = ParameterSpace.from_file(os.path.join(folder, cls.PARAM_SPACE_FILE))
        surr = gp_surrogate.from_saved(folder)
        with open(os.path.join(folder, cls.OPT_ATTRS_FILE)) as fh:
            attrs = json.load(fh)
        opt = cls(parameter_space=space, gp_surrogate=surr, callbacks=callbacks, saver=saver)
        for name, value in attrs.items():
            setattr(opt, name, value)
        assert callable(objective_function)
        opt.obj_func = objective_function
        assert callable(eval_repeats_function)
        opt.eval_repeats_function = partial(eval_repeats_function, axis=0)
        return opt.resume_run(additional_budget=additional_budget), opt

"""Differential evolution on the acquisition function -- robo/maximizers/differential_evolution.py:7-51, i.e.
``maximizer="differential_evolution"`` of the front ends (robo/fmin/bayesian_optimization.py:135-136,
robo/fmin/entropy_search.py:108-109): ``scipy.optimize.differential_evolution(-acq(clip(x)), box, maxiter=n_iters)`` with
SciPy's defaults (population 15 D, best1bin, immediate updating, L-BFGS-B polish; its random numbers come from the global
NumPy stream), infinite values mapped to ``sys.float_info.max`` (:27-34), the result clipped to the box (:51).

In that form every objective call evaluates ONE point (on the device: the matrix-vector form of the explicit-inverse
posterior, 0.05 ms per call at N = 4096).  ``batched=True`` (not in the reference) hands SciPy a vectorised objective
instead: a whole generation -- 15 D trial points -- is scored by ONE device call (SciPy then updates the population once per
generation, "deferred", so the search path differs from the reference's immediate updating; same algorithm family, same
stopping rule)."""
import sys

import numpy as np
from scipy import optimize

from robo_amd.maximizers.random_sampling import BaseMaximizer


class DifferentialEvolution(BaseMaximizer):

    def __init__(self, objective_function, lower, upper, n_iters=20, rng=None, batched=False):
        self.n_iters = n_iters
        self.batched = batched
        super(DifferentialEvolution, self).__init__(objective_function, lower, upper, rng)

    def _score(self, points):
        """-acq at the rows of `points` (clipped to the box), +inf/-inf -> the largest float (as the reference maps them)"""
        values = -np.asarray(self.objective_func(np.clip(points, self.lower, self.upper)), dtype=np.float64).reshape(-1)
        values[np.isinf(values)] = sys.float_info.max
        return values

    def maximize(self):
        box = list(zip(self.lower, self.upper))
        if self.batched:
            # SciPy passes a (D, S) array and expects (S,) values
            found = optimize.differential_evolution(lambda pop: self._score(np.atleast_2d(pop.T)), box, maxiter=self.n_iters,
                                                    vectorized=True, updating="deferred")
        else:
            found = optimize.differential_evolution(lambda x: self._score(x[None, :]), box, maxiter=self.n_iters)
        return np.clip(found["x"], self.lower, self.upper)

"""Differential evolution on the acquisition function -- robo/maximizers/differential_evolution.py:7-51, i.e.
``maximizer="differential_evolution"`` of the front ends (robo/fmin/bayesian_optimization.py:135-136,
robo/fmin/entropy_search.py:108-109): ``scipy.optimize.differential_evolution(-acq(clip(x)), box, maxiter=n_iters)`` with
SciPy's defaults (population 15 D, best1bin, immediate updating, L-BFGS-B polish; its random numbers come from the global
NumPy stream), infinite values mapped to ``sys.float_info.max`` (:27-34), the result clipped to the box (:51).
Every objective call evaluates ONE point (see scipy_optimizer.py)."""
import sys

import numpy as np
from scipy import optimize

from robo_amd.maximizers.random_sampling import BaseMaximizer


class DifferentialEvolution(BaseMaximizer):

    def __init__(self, objective_function, lower, upper, n_iters=20, rng=None):
        self.n_iters = n_iters
        super(DifferentialEvolution, self).__init__(objective_function, lower, upper, rng)

    def _negated(self, x):
        a = -np.asarray(self.objective_func(np.array([np.clip(x, self.lower, self.upper)]))).reshape(-1)
        return sys.float_info.max if np.any(np.isinf(a)) else a

    def maximize(self):
        res = optimize.differential_evolution(self._negated, list(zip(self.lower, self.upper)), maxiter=self.n_iters)
        return np.clip(res["x"], self.lower, self.upper)

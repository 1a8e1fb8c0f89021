"""Multi-start L-BFGS-B maximisation of the acquisition function -- the constructor, the start-point recipe and the
selection rule of robo/maximizers/scipy_optimizer.py:11-84, so that ``maximizer="scipy"`` of the front ends
(robo/fmin/bayesian_optimization.py:133-134, robo/fmin/entropy_search.py:106-107) behaves as it does there:

  starts   ``int(n_restarts / 2)`` uniform points (``init_random_uniform`` WITHOUT a generator: a fresh RandomState seeded
           from the global stream, :63) + ``int(n_restarts / 2)`` draws N(incumbent, 0.5) from the GLOBAL ``np.random`` (:64-66)
  search   ``scipy.optimize.minimize(method="L-BFGS-B", bounds=box)`` on ``-acq(clip(x))`` with finite-difference
           gradients; NaN inputs and infinite values map to ``sys.float_info.max`` (:39-50)
  result   the end point with the lowest value, clipped to the box (:80-82)

Mirrored quirk: the ``rng`` argument is accepted and not handed on (:33-37: the base class draws its own).
Every objective call evaluates ONE point; on the device that is the matrix-vector form of the explicit-inverse posterior
(robo_amd/csrc/winv.hip winv_gemv_kernel: 0.05 ms per call at N = 4096).
"""
import sys

import numpy as np
from scipy import optimize

from robo_amd.initial_design import init_random_uniform
from robo_amd.maximizers.random_sampling import BaseMaximizer


class SciPyOptimizer(BaseMaximizer):

    def __init__(self, objective_function, lower, upper, n_restarts=10, verbosity=False, rng=None):
        self.n_restarts = n_restarts
        self.verbosity = verbosity
        super(SciPyOptimizer, self).__init__(objective_function, lower, upper)      # sic: rng is not passed on

    def _negated(self, x):
        if np.any(np.isnan(x)):
            return sys.float_info.max
        a = -np.asarray(self.objective_func(np.array([np.clip(x, self.lower, self.upper)]))).reshape(-1)[0]
        return sys.float_info.max if np.isinf(a) else float(a)

    def starts(self):
        half = int(self.n_restarts * 0.5)
        uniform = init_random_uniform(self.lower, self.upper, half)
        loc = self.objective_func.model.get_incumbent()[0]
        # one (half, D) draw takes the same numbers from the global stream as the reference's per-row list comprehension
        around = np.random.normal(loc=loc, scale=np.ones([self.lower.shape[0]]) * 0.5, size=(half, self.lower.shape[0]))
        return np.append(uniform, around.reshape(half, self.lower.shape[0]), axis=0)

    def maximize(self):
        ends, values = [], []
        bounds = list(zip(self.lower, self.upper))
        for start in self.starts():
            res = optimize.minimize(self._negated, start, method="L-BFGS-B", bounds=bounds,
                                    options={"disp": self.verbosity})
            ends.append(res["x"])
            values.append(res["fun"])
        return np.clip(ends[int(np.argmin(values))], self.lower, self.upper)

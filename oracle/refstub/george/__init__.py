"""Test-only stand-in for the ``george`` package (TEST INFRASTRUCTURE -- never imported by robo_amd).

Purpose: let the REFERENCE'S OWN classes -- ``robo.models.gaussian_process.GaussianProcess``,
``GaussianProcessMCMC``, ``FabolasGP(MCMC)``, ``robo.acquisition_functions.information_gain*`` and
``robo.fmin.*`` -- import and execute unchanged in the build container, so that golden fixtures are
produced by the reference's code and not by a restatement of it (tests/golden/make_golden_ref.py).

george itself (``requirements.txt:8``, ``git+https://github.com/automl/george.git@development``,
a branch pin with no version) is not vendored under /root/reference and cannot be built here, so
this module restates the slice of its API that RoBO touches (every call site is listed in
SURVEY.md A.1) with george's published algorithm -- standard GP regression on a Cholesky factor:

  george.GP(kernel, mean=float)
  gp.compute(X, yerr)            K = k(X, X) + (yerr^2 + 1.25e-12) I, scipy cho_factor; LinAlgError when not PD
  gp.log_likelihood(y, quiet)    -1/2 (r^T K^-1 r + log|K| + N log 2 pi), r = y - mean
  gp.predict(y, t)               (k(t,X) K^-1 r + mean,  k(t,t) - k(t,X) K^-1 k(X,t))
  gp.sample_conditional(y, t, n)
  gp._compute_alpha(y), gp._alpha, gp._x, gp.solver.apply_inverse(I, in_place=True)

What stays "contract" (unpinnable without george's source): the kernel FORMULAS and their
parameterisation in ``george.kernels``.  Everything the reference's own Python does
around them is now executed, not restated.
"""
import numpy as np
import scipy.linalg as sla

from . import kernels  # noqa: F401

__version__ = "0.0-refstub"

TINY = 1.25e-12


class _Solver(object):
    """george.BasicSolver slice: holds the factor of the last ``compute``."""

    def __init__(self):
        self.computed = False
        self._factor = None
        self.log_determinant = None

    def compute(self, K):
        self._factor = sla.cho_factor(K, lower=True, overwrite_a=True, check_finite=False)
        self.log_determinant = 2.0 * np.sum(np.log(np.diag(self._factor[0])))
        self.computed = True

    def apply_inverse(self, y, in_place=False):
        return sla.cho_solve(self._factor, y, overwrite_b=in_place, check_finite=False)


class GP(object):

    def __init__(self, kernel, mean=None, **kwargs):
        self.kernel = kernel
        self.mean = 0.0 if mean is None else float(mean)
        self.solver = _Solver()
        self._x = None
        self._alpha = None
        self._y = None

    @property
    def computed(self):
        return self.solver.computed

    def compute(self, x, yerr=0.0, **kwargs):
        x = np.atleast_2d(np.asarray(x, dtype=np.float64))
        self._x = x
        self._yerr2 = np.asarray(yerr, dtype=np.float64) ** 2 * np.ones(x.shape[0])
        K = self.kernel.get_value(x)
        K[np.diag_indices_from(K)] += self._yerr2 + TINY
        self._alpha = None
        self.solver.computed = False
        self.solver.compute(K)          # raises numpy.linalg.LinAlgError when K is not PD

    def _compute_alpha(self, y):
        r = np.asarray(y, dtype=np.float64) - self.mean
        self._y = y
        self._alpha = self.solver.apply_inverse(r)
        return self._alpha

    def log_likelihood(self, y, quiet=False):
        if not self.solver.computed:
            raise RuntimeError("compute() first")
        r = np.asarray(y, dtype=np.float64) - self.mean
        ll = -0.5 * (np.dot(r, self.solver.apply_inverse(r)) + self.solver.log_determinant
                     + r.shape[0] * np.log(2.0 * np.pi))
        return ll if np.isfinite(ll) else -np.inf

    lnlikelihood = log_likelihood

    def predict(self, y, t, return_cov=True, return_var=False):
        t = np.atleast_2d(np.asarray(t, dtype=np.float64))
        alpha = self._compute_alpha(y)
        Kxs = self.kernel.get_value(t, self._x)
        mu = np.dot(Kxs, alpha) + self.mean
        cov = self.kernel.get_value(t) - np.dot(Kxs, self.solver.apply_inverse(Kxs.T))
        return mu, cov

    def sample_conditional(self, y, t, size=1):
        mu, cov = self.predict(y, t)
        out = np.random.multivariate_normal(mu, cov, size=size)
        return out[0] if size == 1 else out

"""Expected improvement  EI(x) = s (z Phi(z) + phi(z)),  z = (eta - mu - par) / s.

Semantics of robo/acquisition_functions/ei.py:39-93: incumbent from the model when ``eta``
is None (:67-68); a single zero-sigma point collapses the whole batch to ``[[0]]``
(:72-74); any negative value raises ValueError (:86-88).  The device reports both
conditions in a flags word; this shim turns them into the reference's behaviour.
"""
import logging

import numpy as np

from robo_amd import _lib
from robo_amd.acquisition_functions.base_acquisition import ClosedFormAcquisition

logger = logging.getLogger(__name__)


class EI(ClosedFormAcquisition):
    kind = "ei"

    def __init__(self, model, par=0.0):
        super(EI, self).__init__(model, par)

    def compute(self, X, derivative=False, eta=None, **kwargs):
        f, flags = self._evaluate(X, eta)
        if flags & _lib.FLAG_ZERO_SIGMA:
            f = np.array([[0]])
            return (f, np.zeros((1, X.shape[1]))) if derivative else f      # ei.py:72-74
        if flags & _lib.FLAG_NEGATIVE_EI:
            logger.error("Expected Improvement is smaller than 0!")
            raise ValueError
        if not derivative:
            return f
        # ei.py:80-85:  dEI/dx = -dm/dx Phi(z) + ds/dx phi(z),  ds/dx = dv/dx / (2 s).  The reference takes
        # ``dmdx[0]`` -- it is written for one point (1, D); here every row gets its own gradient, which is the
        # same thing for the (1, D) calls of robo/maximizers/scipy_optimizer.py.
        from scipy.stats import norm
        m, v, dmdx, dvdx = self._moment_gradients(X)
        s = np.sqrt(v)
        z = (self._eta(eta) - m - self.par) / s
        df = -dmdx * norm.cdf(z)[:, None] + (dvdx / (2 * s)[:, None]) * norm.pdf(z)[:, None]
        return f, df

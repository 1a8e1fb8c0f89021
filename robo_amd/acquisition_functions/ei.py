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
        self._no_derivative(derivative)
        f, flags = self._evaluate(X, eta)
        if flags & _lib.FLAG_ZERO_SIGMA:
            return np.array([[0]])
        if flags & _lib.FLAG_NEGATIVE_EI:
            logger.error("Expected Improvement is smaller than 0!")
            raise ValueError
        return f

"""Log expected improvement, branch for branch robo/acquisition_functions/log_ei.py:35-122
(degenerate cases f_min == mu, sigma == 0, and -inf when a >= b), evaluated for the whole
batch by the device kernel instead of the reference's Python loop over points (:79-120).
"""
import logging

from robo_amd.acquisition_functions.base_acquisition import ClosedFormAcquisition

logger = logging.getLogger(__name__)


class LogEI(ClosedFormAcquisition):
    kind = "log_ei"

    def __init__(self, model, par=0.0, **kwargs):
        super(LogEI, self).__init__(model, par)

    def compute(self, X, derivative=False, eta=None, **kwargs):
        if derivative:
            logger.error("LogEI does not support derivative calculation until now")
            return
        f, _ = self._evaluate(X, eta)
        return f

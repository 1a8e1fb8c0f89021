"""Lower confidence bound  LCB(x) = -(mu - par * sqrt(var)),  par = kappa = 1 by default
(robo/acquisition_functions/lcb.py:12,40-71; RoBO maximises, hence the sign)."""
from robo_amd.acquisition_functions.base_acquisition import ClosedFormAcquisition


class LCB(ClosedFormAcquisition):
    kind = "lcb"
    needs_eta = False

    def __init__(self, model, par=1.0):
        super(LCB, self).__init__(model, par)

    def compute(self, X, derivative=False, **kwargs):
        self._no_derivative(derivative)
        f, _ = self._evaluate(X, None)
        return f

"""Lower confidence bound  LCB(x) = -(mu - par * sqrt(var)),  par = kappa = 1 by default
(robo/acquisition_functions/lcb.py:12,40-71; RoBO maximises, hence the sign)."""
import numpy as np

from robo_amd.acquisition_functions.base_acquisition import ClosedFormAcquisition


class LCB(ClosedFormAcquisition):
    kind = "lcb"
    needs_eta = False

    def __init__(self, model, par=1.0):
        super(LCB, self).__init__(model, par)

    def compute(self, X, derivative=False, **kwargs):
        f, _ = self._evaluate(X, None)
        if not derivative:
            return f
        # lcb.py:66-68:  grad = -(dm - par dv / (2 sqrt(var))), as (M, D) (the reference subtracts a (M, D)
        # array from the (M, D, 1) mean gradient, which broadcasts to (M, D, D): a shape slip of a path that
        # never ran, since no reference model implements predictive_gradients)
        m, v, dmdx, dvdx = self._moment_gradients(X)
        return f, -(dmdx - self.par * dvdx / (2 * np.sqrt(v))[:, None])

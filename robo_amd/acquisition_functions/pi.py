"""Probability of improvement  PI(x) = Phi((eta - mu - par) / s)
(robo/acquisition_functions/pi.py:34-73; eta is always the model's incumbent, :59)."""
import numpy as np

from robo_amd.acquisition_functions.base_acquisition import ClosedFormAcquisition


class PI(ClosedFormAcquisition):
    kind = "pi"

    def __init__(self, model, par=0.0):
        super(PI, self).__init__(model, par)

    def compute(self, X_test, derivative=False, **kwargs):
        f, _ = self._evaluate(X_test, kwargs.get("eta"))
        if not derivative:
            return f
        # pi.py:65-71:  dPI/dx = -(phi(z) / s) (dm/dx + ds/dx z), row-wise (the reference indexes point 0)
        from scipy.stats import norm
        m, v, dmdx, dvdx = self._moment_gradients(X_test)
        s = np.sqrt(v)
        z = (self._eta(kwargs.get("eta")) - m - self.par) / s
        df = (-norm.pdf(z) / s)[:, None] * (dmdx + (dvdx / (2 * s)[:, None]) * z[:, None])
        return f, df

"""Probability of improvement  PI(x) = Phi((eta - mu - par) / s)
(robo/acquisition_functions/pi.py:34-73; eta is always the model's incumbent, :59)."""
from robo_amd.acquisition_functions.base_acquisition import ClosedFormAcquisition


class PI(ClosedFormAcquisition):
    kind = "pi"

    def __init__(self, model, par=0.0):
        super(PI, self).__init__(model, par)

    def compute(self, X_test, derivative=False, **kwargs):
        self._no_derivative(derivative)
        f, _ = self._evaluate(X_test, kwargs.get("eta"))
        return f

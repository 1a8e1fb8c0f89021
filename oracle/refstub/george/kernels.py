"""``george.kernels`` slice used by RoBO (TEST INFRASTRUCTURE; see the package docstring).

Generic composition, deliberately NOT built on oracle/gp_oracle.kernel_matrix: every elementary
kernel evaluates its own formula on its own ``axes`` and ``Product`` multiplies, so that
tests/test_oracle.py can cross-check the (kind, theta) restatement in the oracle against an
independently written evaluation of the same contract:

* parameter vectors are in LOG space, in product order (left factor first);
* ``Matern52Kernel(metric, ndim, axes)``: k = (1 + sqrt(5 r2) + 5 r2 / 3) exp(-sqrt(5 r2)),
  r2 = sum_{d in axes} (x_d - x'_d)^2 / m_d, the metric m_d is a SQUARED length scale;
* ``ExpSquaredKernel``: k = exp(-r2 / 2);
* ``b * kernel`` for a Python scalar b: ``ConstantKernel(log_constant = log(b / ndim))``;
* ``BayesianLinearRegressionKernel(log_a, log_b, ndim, axes)``: e^{log_a} + e^{log_b} u u'
  (degree-1 Bayesian linear regression, the Fabolas paper's kernel; defined choice).
"""
import numpy as np

__all__ = ["Kernel", "ConstantKernel", "Product", "Matern52Kernel", "ExpSquaredKernel",
           "BayesianLinearRegressionKernel"]


class Kernel(object):
    ndim = 1

    # ---- modelling protocol slice ------------------------------------------------------------
    def __len__(self):
        return self.get_parameter_vector().shape[0]

    def get_parameter_vector(self):
        raise NotImplementedError

    def set_parameter_vector(self, v):
        raise NotImplementedError

    @property
    def vector(self):
        return self.get_parameter_vector()

    @vector.setter
    def vector(self, v):
        self.set_parameter_vector(v)

    @property
    def pars(self):
        """george-0.2 style linear-space parameters (fabolas_gp.py:58 reads ``len(kernel.pars)``)."""
        return np.exp(self.get_parameter_vector())

    def __getitem__(self, k):
        return self.get_parameter_vector()[k]

    def __setitem__(self, k, v):
        vec = self.get_parameter_vector()
        vec[k] = v
        self.set_parameter_vector(vec)

    # ---- algebra ---------------------------------------------------------------------------------
    def __mul__(self, b):
        if not hasattr(b, "get_value"):
            return Product(ConstantKernel(log_constant=np.log(float(b) / self.ndim), ndim=self.ndim), self)
        return Product(self, b)

    def __rmul__(self, b):
        if not hasattr(b, "get_value"):
            return Product(ConstantKernel(log_constant=np.log(float(b) / self.ndim), ndim=self.ndim), self)
        return Product(b, self)

    # ---- evaluation -----------------------------------------------------------------------------
    def get_value(self, x1, x2=None):
        x1 = np.atleast_2d(np.asarray(x1, dtype=np.float64))
        x2 = x1 if x2 is None else np.atleast_2d(np.asarray(x2, dtype=np.float64))
        return self._value(x1, x2)

    value = get_value

    def gradient(self, x1, x2=None):
        """(N1, N2, P): derivative w.r.t. every entry of the log-space parameter vector."""
        x1 = np.atleast_2d(np.asarray(x1, dtype=np.float64))
        x2 = x1 if x2 is None else np.atleast_2d(np.asarray(x2, dtype=np.float64))
        return self._gradient(x1, x2)

    get_gradient = gradient


class ConstantKernel(Kernel):

    def __init__(self, log_constant, ndim=1, axes=None):
        self.ndim = int(ndim)
        self.log_constant = float(log_constant)

    def get_parameter_vector(self):
        return np.array([self.log_constant])

    def set_parameter_vector(self, v):
        self.log_constant = float(np.asarray(v).ravel()[0])

    def _value(self, x1, x2):
        return np.full((x1.shape[0], x2.shape[0]), np.exp(self.log_constant))

    def _gradient(self, x1, x2):
        return self._value(x1, x2)[:, :, None]


class Product(Kernel):

    def __init__(self, k1, k2):
        assert k1.ndim == k2.ndim, "Dimension mismatch"
        self.k1, self.k2 = k1, k2
        self.ndim = k1.ndim

    def get_parameter_vector(self):
        return np.append(self.k1.get_parameter_vector(), self.k2.get_parameter_vector())

    def set_parameter_vector(self, v):
        v = np.asarray(v, dtype=np.float64)
        n1 = len(self.k1)
        assert v.shape[0] == n1 + len(self.k2)
        self.k1.set_parameter_vector(v[:n1])
        self.k2.set_parameter_vector(v[n1:])

    def _value(self, x1, x2):
        return self.k1._value(x1, x2) * self.k2._value(x1, x2)

    def _gradient(self, x1, x2):
        v1, v2 = self.k1._value(x1, x2), self.k2._value(x1, x2)
        return np.concatenate((self.k1._gradient(x1, x2) * v2[:, :, None],
                               self.k2._gradient(x1, x2) * v1[:, :, None]), axis=2)


class _Radial(Kernel):
    """stationary kernel on an axis-aligned metric (one squared length scale per axis, or one for all)"""

    def __init__(self, metric, ndim=1, axes=None):
        self.ndim = int(ndim)
        if axes is None:
            self.axes = np.arange(self.ndim)
        else:
            self.axes = np.atleast_1d(np.asarray(axes, dtype=int))
        metric = np.atleast_1d(np.asarray(metric, dtype=np.float64))
        if metric.shape[0] not in (1, self.axes.shape[0]):
            raise ValueError("Dimension mismatch")
        self.log_metric = np.log(metric)

    def get_parameter_vector(self):
        return self.log_metric.copy()

    def set_parameter_vector(self, v):
        v = np.atleast_1d(np.asarray(v, dtype=np.float64))
        assert v.shape == self.log_metric.shape
        self.log_metric = v.copy()

    def _metric_per_axis(self):
        m = np.exp(self.log_metric)
        return np.full(self.axes.shape[0], m[0]) if m.shape[0] == 1 else m

    def _sq(self, x1, x2):
        """list over axes of (x1_d - x2_d)^2 / m_d"""
        m = self._metric_per_axis()
        return [np.subtract.outer(x1[:, d], x2[:, d]) ** 2 / m[i] for i, d in enumerate(self.axes)]

    def _value(self, x1, x2):
        return self._f(sum(self._sq(x1, x2)))

    def _gradient(self, x1, x2):
        sq = self._sq(x1, x2)
        df = self._df(sum(sq))
        # d r2 / d log m_d = -(x_d - x'_d)^2 / m_d
        if self.log_metric.shape[0] == 1:
            return (df * -sum(sq))[:, :, None]
        return np.stack([df * -s for s in sq], axis=2)


class Matern52Kernel(_Radial):

    @staticmethod
    def _f(r2):
        s = np.sqrt(5.0 * r2)
        return (1.0 + s + 5.0 * r2 / 3.0) * np.exp(-s)

    @staticmethod
    def _df(r2):
        s = np.sqrt(5.0 * r2)
        return -(5.0 / 6.0) * (1.0 + s) * np.exp(-s)


class ExpSquaredKernel(_Radial):

    @staticmethod
    def _f(r2):
        return np.exp(-0.5 * r2)

    @staticmethod
    def _df(r2):
        return -0.5 * np.exp(-0.5 * r2)


class BayesianLinearRegressionKernel(Kernel):

    def __init__(self, log_a=0.0, log_b=0.0, ndim=1, axes=None):
        self.ndim = int(ndim)
        self.axis = int(np.atleast_1d(0 if axes is None else axes)[0])
        self.log_a, self.log_b = float(log_a), float(log_b)

    def get_parameter_vector(self):
        return np.array([self.log_a, self.log_b])

    def set_parameter_vector(self, v):
        self.log_a, self.log_b = float(v[0]), float(v[1])

    def _value(self, x1, x2):
        return np.exp(self.log_a) + np.exp(self.log_b) * np.multiply.outer(x1[:, self.axis], x2[:, self.axis])

    def _gradient(self, x1, x2):
        uu = np.multiply.outer(x1[:, self.axis], x2[:, self.axis])
        return np.stack([np.full(uu.shape, np.exp(self.log_a)), np.exp(self.log_b) * uu], axis=2)

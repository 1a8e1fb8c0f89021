"""NumPy/SciPy restatement of the reference's GP-posterior + acquisition path.

TEST INFRASTRUCTURE ONLY -- never imported by the product (``robo_amd``).

What is restated, and from where (paths relative to /root/reference):

* george calls made by ``robo/models/gaussian_process.py`` -- ``gp.compute``
  (:119,155), ``gp.log_likelihood`` (:159), ``gp.predict`` (:280).  george is a
  third-party dependency (``requirements.txt:8``,
  ``git+https://github.com/automl/george.git@development``, branch pin, no
  version) whose source is NOT in the reference tree and which cannot be built
  here.  Its published algorithm (standard GP regression on a Cholesky factor)
  is restated below.  **PARITY UNPINNED for the kernel values**: nothing in the
  reference tree pins the Matern-5/2 parameterisation or george's jitter; the
  definitions in ``kernel_matrix``/``JITTER`` are this project's contract
  (SURVEY.md A.2).  The posterior *algebra* is pinned by the reference's own
  test identity (test/test_models/test_gaussian_process.py:44-49), re-checked
  in tests/test_oracle.py.  As an independent third-party check (not a pin to
  george) the same test file compares kernel matrix, posterior, log-likelihood
  and likelihood gradient with scikit-learn's Matern(nu=2.5)/RBF +
  GaussianProcessRegressor: equal to rounding.
* ``GaussianProcess.train/nll/predict/get_incumbent``
  (robo/models/gaussian_process.py:70-124,129-166,251-296,334-352) ->
  :class:`OracleGP`.
* ``GaussianProcess.grad_nll`` (robo/models/gaussian_process.py:168-191) ->
  :func:`gp_grad_log_likelihood` (+ :func:`kernel_gradient` for george's
  ``kernel.gradient``, PARITY UNPINNED like the kernel values).
* ``GaussianProcessMCMC.predict`` mixture (robo/models/gaussian_process_mcmc.py:230-247)
  -> :func:`mcmc_mixture`.
* ``EI/LogEI/PI/LCB.compute`` (robo/acquisition_functions/ei.py:65-88,
  log_ei.py:74-120, pi.py:57-63, lcb.py:62-65) -> :func:`ei`, :func:`log_ei`,
  :func:`pi`, :func:`lcb`.  **PINNED**: tests/golden/make_golden.py runs the
  reference's own classes (importable here) on the same (mean, var) and the
  committed fixtures hold their outputs; tests/test_oracle.py compares.
* ``MarginalizationGPMCMC.compute`` (robo/acquisition_functions/marginalization.py:115-121)
  -> :func:`marginalize`.
* ``zero_one_normalization`` etc. (robo/util/normalization.py:4-32).
"""
import numpy as np
import scipy.linalg as sla
from scipy.special import erfc, erfcx

__all__ = [
    "JITTER", "EPS", "kernel_matrix", "kernel_diag", "n_kernel_params", "gp_compute",
    "gp_log_likelihood", "gp_grad_log_likelihood", "kernel_gradient", "kernel_input_gradient",
    "gp_predictive_gradients", "gp_predict", "gp_predict_diag", "OracleGP", "mcmc_mixture",
    "norm_cdf", "norm_pdf", "norm_logpdf", "norm_logcdf",
    "ei", "log_ei", "pi", "lcb", "marginalize", "np_argmax",
    "zero_one_normalization", "zero_one_unnormalization",
    "zero_mean_unit_var_normalization", "zero_mean_unit_var_unnormalization",
]

#: george adds this to the diagonal on top of yerr**2 (unverified, contract).
JITTER = 1.25e-12
#: variance floor, robo/models/gaussian_process.py:291-294 (np.finfo(float64).eps)
EPS = float(np.finfo(np.float64).eps)


# --------------------------------------------------------------------------
# robo/util/normalization.py:4-32
# --------------------------------------------------------------------------
def zero_one_normalization(X, lower=None, upper=None):
    if lower is None:
        lower = np.min(X, axis=0)
    if upper is None:
        upper = np.max(X, axis=0)
    return np.true_divide((X - lower), (upper - lower)), lower, upper


def zero_one_unnormalization(Xn, lower, upper):
    return lower + (upper - lower) * Xn


def zero_mean_unit_var_normalization(X, mean=None, std=None):
    if mean is None:
        mean = np.mean(X, axis=0)
    if std is None:
        std = np.std(X, axis=0)
    return (X - mean) / std, mean, std


def zero_mean_unit_var_unnormalization(Xn, mean, std):
    return Xn * std + mean


# --------------------------------------------------------------------------
# george kernels (contract: SURVEY.md A.2)
# theta_k = [log amp, log m_1 .. log m_D]  (m_d = SQUARED length scale)
# --------------------------------------------------------------------------
def n_kernel_params(kind, D):
    """number of kernel parameters (without the noise) for D input columns"""
    if kind in ("matern52", "rbf"):
        return 1 + D
    if kind == "fabolas":
        return 1 + (D - 1) + 2
    raise ValueError(kind)


def _matern52_unit(r2):
    s = np.sqrt(r2.dtype.type(5.0) * r2)
    return (r2.dtype.type(1.0) + s + r2.dtype.type(5.0) * r2 / r2.dtype.type(3.0)) * np.exp(-s)


def _r2(metric, X1, X2, dtype):
    m = np.asarray(metric, dtype=np.float64)
    # scaling in fp64 (the device pre-scales the inputs once in fp64), differences in `dtype`
    A = (X1 / np.sqrt(m)).astype(dtype)
    B = (X2 / np.sqrt(m)).astype(dtype)
    # direct differences (not the |a|^2+|b|^2-2ab expansion): exact zero on the
    # diagonal and no cancellation, like george's metric evaluation.
    r2 = np.zeros((A.shape[0], B.shape[0]), dtype=dtype)
    for d in range(A.shape[1]):
        diff = A[:, d][:, None] - B[:, d][None, :]
        r2 += diff * diff
    return r2


def kernel_matrix(kind, theta_k, X1, X2=None, dtype=np.float64):
    """k(X1, X2) for ``amp * Matern52Kernel(metric, ndim=D)`` / ``amp * ExpSquaredKernel`` and the
    Fabolas product kernel (contract: SURVEY.md A.2).

    Call sites restated: robo/fmin/bayesian_optimization.py:75-81 (construction),
    robo/fmin/fabolas.py:104-117, test/test_models/test_gaussian_process.py:44-46
    (``kernel.get_value``).  ``dtype=np.float32`` is the mixed-precision K-build of BASELINE
    config 5: covariance entries evaluated in fp32, returned widened to fp64.
    """
    X1 = np.asarray(X1, dtype=np.float64)
    X2 = X1 if X2 is None else np.asarray(X2, dtype=np.float64)
    T = np.dtype(dtype).type
    amp = T(np.exp(theta_k[0]))
    if kind == "fabolas":
        # theta_k = [log amp, log m_1..m_D, log_a, log_b]; columns 0..D-1 inputs, column D = basis(s)
        D = X1.shape[1] - 1
        m = np.exp(np.asarray(theta_k[1:1 + D], dtype=np.float64))
        a, b = T(np.exp(theta_k[1 + D])), T(np.exp(theta_k[2 + D]))
        A = (X1[:, :D] / np.sqrt(m)).astype(dtype)
        B = (X2[:, :D] / np.sqrt(m)).astype(dtype)
        prod = np.ones((X1.shape[0], X2.shape[0]), dtype=dtype)
        for d in range(D):
            diff = A[:, d][:, None] - B[:, d][None, :]
            prod *= _matern52_unit(diff * diff)
        uu = X1[:, D].astype(dtype)[:, None] * X2[:, D].astype(dtype)[None, :]
        return (amp * prod * (a + b * uu)).astype(np.float64)
    r2 = _r2(np.exp(np.asarray(theta_k[1:], dtype=np.float64)), X1, X2, dtype)
    if kind == "matern52":
        return (amp * _matern52_unit(r2)).astype(np.float64)
    if kind == "rbf":
        return (amp * np.exp(T(-0.5) * r2)).astype(np.float64)
    raise ValueError(kind)


def kernel_diag(kind, theta_k, X):
    """k(x, x): amp for the stationary kernels, amp (a + b u^2) for the Fabolas kernel."""
    amp = np.exp(theta_k[0])
    if kind == "fabolas":
        D = X.shape[1] - 1
        return amp * (np.exp(theta_k[1 + D]) + np.exp(theta_k[2 + D]) * X[:, D] ** 2)
    return np.full(X.shape[0], amp)


# --------------------------------------------------------------------------
# george.GP.compute / log_likelihood / predict
# --------------------------------------------------------------------------
def gp_compute(kind, theta, X, dtype=np.float64):
    """``gp.compute(X, yerr=sqrt(sigma2))``: K + (sigma2 + JITTER) I, Cholesky.

    theta = [theta_k..., log sigma2] (robo/models/gaussian_process.py:151-155).
    Raises np.linalg.LinAlgError when not PD (what :120,:156 catch).
    Returns the lower factor L.
    """
    theta = np.asarray(theta, dtype=np.float64)
    K = kernel_matrix(kind, theta[:-1], X, dtype=dtype)
    K[np.diag_indices_from(K)] += np.exp(theta[-1]) + JITTER
    return sla.cholesky(K, lower=True, check_finite=False)


def kernel_gradient(kind, theta_k, X):
    """``kernel.gradient(X)`` -> (N, N, P_k): derivative of k(X, X) w.r.t. every entry of the
    LOG-space parameter vector (call site robo/models/gaussian_process.py:181).  george's own
    formulas are not in the tree (PARITY UNPINNED, like the kernel values); these are the exact
    derivatives of :func:`kernel_matrix`, checked against central differences of it in
    tests/test_oracle.py.
    """
    X = np.asarray(X, dtype=np.float64)
    K = kernel_matrix(kind, theta_k, X)
    N = X.shape[0]
    G = np.zeros((N, N, len(theta_k)))
    G[:, :, 0] = K                                        # d/d log amp
    if kind == "fabolas":
        D = X.shape[1] - 1
        m = np.exp(np.asarray(theta_k[1:1 + D], dtype=np.float64))
        a, b = np.exp(theta_k[1 + D]), np.exp(theta_k[2 + D])
        for d in range(D):
            diff = (X[:, d][:, None] - X[:, d][None, :]) / np.sqrt(m[d])
            s = diff * diff
            t = np.sqrt(5.0 * s)
            # d log matern52(s) / d log m = -(f'/f) s,  f'/f = -(5/6)(1 + t) / (1 + t + 5 s / 3)
            G[:, :, 1 + d] = K * (5.0 / 6.0) * (1.0 + t) * s / (1.0 + t + 5.0 * s / 3.0)
        uu = X[:, D][:, None] * X[:, D][None, :]
        B = a + b * uu
        G[:, :, 1 + D] = K * a / B
        G[:, :, 2 + D] = K * b * uu / B
        return G
    amp = np.exp(theta_k[0])
    m = np.exp(np.asarray(theta_k[1:], dtype=np.float64))
    r2 = _r2(m, X, X, np.float64)
    if kind == "matern52":
        t = np.sqrt(5.0 * r2)
        dk = -amp * (5.0 / 6.0) * (1.0 + t) * np.exp(-t)   # amp f'(r2)
    elif kind == "rbf":
        dk = -0.5 * K
    else:
        raise ValueError(kind)
    for d in range(X.shape[1]):
        diff = (X[:, d][:, None] - X[:, d][None, :]) / np.sqrt(m[d])
        G[:, :, 1 + d] = dk * (-(diff * diff))             # d r2 / d log m_d = -diff^2
    return G


def gp_grad_log_likelihood(kind, theta, X, y, mean):
    """The likelihood part of ``GaussianProcess.grad_nll`` (robo/models/gaussian_process.py:168-191)
    with its sign flipped: 0.5 einsum('ijk,ij', Kg, alpha alpha^T - K^-1).  As in the reference
    (:178-182) the 'gradient' of the Gram matrix w.r.t. the last entry (log sigma^2) is the IDENTITY,
    not sigma^2 I -- mirrored, see DESIGN.md "Mirrored quirks"."""
    theta = np.asarray(theta, dtype=np.float64)
    L = gp_compute(kind, theta, X)
    r = np.asarray(y, dtype=np.float64) - mean
    alpha = sla.cho_solve((L, True), r, check_finite=False)
    K_inv = sla.cho_solve((L, True), np.eye(L.shape[0]), check_finite=False)
    Kg = kernel_gradient(kind, theta[:-1], X)
    Kg = np.concatenate((Kg, np.eye(L.shape[0])[:, :, None]), axis=2)
    A = np.outer(alpha, alpha) - K_inv
    return 0.5 * np.einsum('ijk,ij', Kg, A)


def gp_log_likelihood(L, y, mean):
    """``gp.log_likelihood(y, quiet=True)`` (robo/models/gaussian_process.py:159)."""
    r = y - mean
    z = sla.solve_triangular(L, r, lower=True, check_finite=False)
    n = L.shape[0]
    return -0.5 * (z @ z + 2.0 * np.sum(np.log(np.diag(L))) + n * np.log(2.0 * np.pi))


def gp_predict(kind, theta, L, X, y, mean, Xs):
    """``gp.predict(y, Xs)`` -> (mu (M,), cov (M,M)); robo/models/gaussian_process.py:280.

    This is the reference's call sequence: FULL covariance (the caller takes
    np.diag afterwards, :286).  Used for full_cov and for the cpu_baseline.
    """
    r = y - mean
    alpha = sla.cho_solve((L, True), r, check_finite=False)
    Kxs = kernel_matrix(kind, theta[:-1], Xs, X)
    mu = Kxs @ alpha + mean
    KinvKxsT = sla.cho_solve((L, True), Kxs.T, check_finite=False)
    cov = kernel_matrix(kind, theta[:-1], Xs) - Kxs @ KinvKxsT
    return mu, cov


def gp_predict_diag(kind, theta, L, X, y, mean, Xs, chunk=4096, dtype=np.float64):
    """Diagonal-only variant of :func:`gp_predict` ("fair" CPU baseline; same math).

    var_c = k(x_c,x_c) - |L^{-1} k(X,x_c)|^2 ;  mu_c = k(x_c,X) alpha + mean.
    """
    r = y - mean
    alpha = sla.cho_solve((L, True), r, check_finite=False)
    M = Xs.shape[0]
    mu = np.empty(M)
    var = np.empty(M)
    for s in range(0, M, chunk):
        e = min(M, s + chunk)
        Kxs = kernel_matrix(kind, theta[:-1], Xs[s:e], X, dtype=dtype)
        mu[s:e] = Kxs @ alpha + mean
        V = sla.solve_triangular(L, Kxs.T, lower=True, check_finite=False)
        var[s:e] = kernel_diag(kind, theta[:-1], Xs[s:e]) - np.sum(V * V, axis=0)
    return mu, var


# --------------------------------------------------------------------------
# robo/models/gaussian_process.py -> OracleGP
# --------------------------------------------------------------------------
class OracleGP(object):
    """Restatement of ``GaussianProcess`` with an explicit (kind, theta)."""

    def __init__(self, kind, theta, prior=None, normalize_output=False,
                 normalize_input=True, lower=None, upper=None, dtype=np.float64):
        self.dtype = dtype
        self.kind = kind
        self.theta = np.asarray(theta, dtype=np.float64).copy()
        self.prior = prior
        self.normalize_output = normalize_output
        self.normalize_input = normalize_input
        self.lower = lower
        self.upper = upper
        self.is_trained = False
        self.models = None

    @property
    def noise(self):
        return float(np.exp(self.theta[-1]))

    # gaussian_process.py:70-124 (do_optimize=False branch; the optimiser is host code)
    def train(self, X, y, do_optimize=False):
        assert X.shape[0] == y.shape[0] and X.ndim == 2 and y.ndim == 1
        if self.normalize_input:
            self.X, self.lower, self.upper = zero_one_normalization(X, self.lower, self.upper)
        else:
            self.X = X
        if self.normalize_output:
            self.y, self.y_mean, self.y_std = zero_mean_unit_var_normalization(y)
            if self.y_std == 0:
                raise ValueError("Cannot normalize output. All targets have the same value")
        else:
            self.y = y
        self.mean = np.mean(self.y, axis=0)
        try:
            self.L = gp_compute(self.kind, self.theta, self.X, self.dtype)
        except np.linalg.LinAlgError:
            # gaussian_process.py:120-122
            self.theta[-1] = np.log(np.exp(self.theta[-1]) * 10)
            self.L = gp_compute(self.kind, self.theta, self.X, self.dtype)
        self.is_trained = True

    # gaussian_process.py:129-166
    def nll(self, theta):
        theta = np.asarray(theta, dtype=np.float64)
        if np.any((-20 > theta) + (theta > 20)):
            return 1e25
        try:
            L = gp_compute(self.kind, theta, self.X, self.dtype)
        except np.linalg.LinAlgError:
            return 1e25
        ll = gp_log_likelihood(L, self.y, self.mean)
        if self.prior is not None:
            ll += self.prior.lnprob(theta)
        return -ll if np.isfinite(ll) else 1e25

    # gaussian_process_mcmc.py:168-202 (same fit, -inf protocol)
    def loglikelihood(self, theta):
        theta = np.asarray(theta, dtype=np.float64)
        if np.any((-20 > theta) + (theta > 20)):
            return -np.inf
        try:
            L = gp_compute(self.kind, theta, self.X, self.dtype)
        except Exception:
            return -np.inf
        ll = gp_log_likelihood(L, self.y, self.mean)
        if self.prior is not None:
            return self.prior.lnprob(theta) + ll
        return ll

    # gaussian_process.py:251-296
    def predict(self, X_test, full_cov=False, diag_only=False, **kwargs):
        assert X_test.ndim == 2
        if not self.is_trained:
            raise Exception('Model has to be trained first!')
        if self.normalize_input:
            Xt, _, _ = zero_one_normalization(X_test, self.lower, self.upper)
        else:
            Xt = X_test
        if diag_only and not full_cov:
            mu, var = gp_predict_diag(self.kind, self.theta, self.L, self.X, self.y, self.mean, Xt,
                                      dtype=self.dtype)
        else:
            mu, var = gp_predict(self.kind, self.theta, self.L, self.X, self.y, self.mean, Xt)
        if self.normalize_output:
            mu = zero_mean_unit_var_unnormalization(mu, self.y_mean, self.y_std)
            var = var * self.y_std ** 2
        if not full_cov and var.ndim == 2:
            var = np.diag(var)
        var = np.clip(var, EPS, np.inf)   # :290-294 (the ':294' line is a no-op after the clip)
        return mu, var

    def predictive_gradients(self, X_test):
        """(dmdx (M, D, 1), dvdx (M, D)) in the caller's input space, GPy convention (see gp_predictive_gradients)"""
        if self.normalize_input:
            Xt, _, _ = zero_one_normalization(X_test, self.lower, self.upper)
            scale = 1.0 / (np.asarray(self.upper, dtype=np.float64) - np.asarray(self.lower, dtype=np.float64))
        else:
            Xt, scale = X_test, 1.0
        dm, dv = gp_predictive_gradients(self.kind, self.theta, self.L, self.X, self.y, self.mean, Xt)
        if self.normalize_output:
            dm, dv = dm * self.y_std, dv * self.y_std ** 2
        return (dm * scale)[:, :, None], dv * scale

    # gaussian_process.py:334-352 / base_model.py:94-106
    def get_incumbent(self):
        best = np.argmin(self.y)
        inc, inc_value = self.X[best], self.y[best]
        if self.normalize_input:
            inc = zero_one_unnormalization(inc, self.lower, self.upper)
        if self.normalize_output:
            inc_value = zero_mean_unit_var_unnormalization(inc_value, self.y_mean, self.y_std)
        return inc, inc_value


def kernel_input_gradient(kind, theta_k, x, X):
    """d k(x, X_n) / d x_d for one test point x (D,) against the rows of X -> (N, D), plus d k(x, x) / d x (D,).
    Exact derivatives of :func:`kernel_matrix` (PARITY UNPINNED like the kernel values); pinned against central
    differences of kernel_matrix in tests/test_oracle.py."""
    x = np.asarray(x, dtype=np.float64)
    X = np.asarray(X, dtype=np.float64)
    amp = np.exp(theta_k[0])
    if kind == "fabolas":
        D = X.shape[1] - 1
        m = np.exp(np.asarray(theta_k[1:1 + D], dtype=np.float64))
        a, b = np.exp(theta_k[1 + D]), np.exp(theta_k[2 + D])
        diff = (x[None, :D] - X[:, :D]) / np.sqrt(m)
        s2 = diff * diff
        t = np.sqrt(5.0 * s2)
        f = (1.0 + t + 5.0 * s2 / 3.0) * np.exp(-t)
        prod = np.prod(f, axis=1)
        lin = a + b * x[D] * X[:, D]
        k = amp * prod * lin
        G = np.empty((X.shape[0], D + 1))
        # f'(s2)/f(s2) = -(5/6)(1 + t)/(1 + t + 5 s2/3); d s2 / d x_d = 2 diff / sqrt(m_d)
        G[:, :D] = k[:, None] * (-(5.0 / 6.0) * (1.0 + t) / (1.0 + t + 5.0 * s2 / 3.0)) * 2.0 * diff / np.sqrt(m)
        G[:, D] = amp * prod * b * X[:, D]
        dself = np.zeros(D + 1)
        dself[D] = 2.0 * amp * b * x[D]
        return G, dself
    m = np.exp(np.asarray(theta_k[1:], dtype=np.float64))
    diff = (x[None, :] - X) / np.sqrt(m)
    r2 = np.sum(diff * diff, axis=1)
    if kind == "matern52":
        t = np.sqrt(5.0 * r2)
        dk = -amp * (5.0 / 6.0) * (1.0 + t) * np.exp(-t)
    elif kind == "rbf":
        dk = -0.5 * amp * np.exp(-0.5 * r2)
    else:
        raise ValueError(kind)
    return dk[:, None] * 2.0 * diff / np.sqrt(m), np.zeros(X.shape[1])


def gp_predictive_gradients(kind, theta, L, X, y, mean, Xs):
    """d mean / d x and d var / d x of the (un-normalised-output) posterior at the rows of Xs -> (M, D), (M, D):
    d mu = (d k_*)^T alpha,  d var = d k(x,x) - 2 (d k_*)^T K^-1 k_*   -- what ``model.predictive_gradients`` has
    to supply to robo/acquisition_functions/ei.py:80-85 (no reference model implements it; the pin is central
    differences of :func:`gp_predict_diag`, tests/test_oracle.py)."""
    theta = np.asarray(theta, dtype=np.float64)
    r = y - mean
    alpha = sla.cho_solve((L, True), r, check_finite=False)
    dm = np.empty(Xs.shape)
    dv = np.empty(Xs.shape)
    for i, x in enumerate(Xs):
        G, dself = kernel_input_gradient(kind, theta[:-1], x, X)
        ks = kernel_matrix(kind, theta[:-1], x[None, :], X)[0]
        beta = sla.cho_solve((L, True), ks, check_finite=False)
        dm[i] = G.T @ alpha
        dv[i] = dself - 2.0 * (G.T @ beta)
    return dm, dv


def mcmc_mixture(mu, var):
    """GaussianProcessMCMC.predict mixture, gaussian_process_mcmc.py:235-247.

    mu, var: (S, M) per-sample predictions -> (m (M,), v (M,)).
    """
    m = mu.mean(axis=0)
    v = np.var(mu, axis=0) + np.mean(var, axis=0)
    v = np.clip(v, EPS, np.inf)
    return m, v


# --------------------------------------------------------------------------
# scipy.stats.norm pieces, written the way the device code computes them so the
# formulas (not only the values) are pinned by tests/test_oracle.py against scipy.
# --------------------------------------------------------------------------
_SQRT1_2 = 0.70710678118654752440
_LOG_SQRT_2PI = 0.91893853320467274178
_SQRT_2PI = 2.50662827463100050242


def norm_cdf(z):
    return 0.5 * erfc(-np.asarray(z, dtype=np.float64) * _SQRT1_2)


def norm_pdf(z):
    z = np.asarray(z, dtype=np.float64)
    return np.exp(-z * z / 2.0) / _SQRT_2PI   # scipy _norm_pdf: exp(-x**2/2)/sqrt(2*pi)


def norm_logpdf(z):
    z = np.asarray(z, dtype=np.float64)
    return -z * z / 2.0 - _LOG_SQRT_2PI      # scipy _norm_logpdf


def norm_logcdf(z):
    z = np.asarray(z, dtype=np.float64)
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        lo = np.log(0.5 * erfcx(-z * _SQRT1_2)) - 0.5 * z * z
        hi = np.log1p(-0.5 * erfc(z * _SQRT1_2))
    return np.where(z < -1.0, lo, hi)


# --------------------------------------------------------------------------
# acquisition functions on (mean, var, eta)
# --------------------------------------------------------------------------
def ei(m, v, eta, par=0.0):
    """robo/acquisition_functions/ei.py:65-88 (batch collapses to [[0]] if any s==0;
    ValueError if any EI<0)."""
    s = np.sqrt(v)
    if (s == 0).any():
        return np.array([[0]])
    z = (eta - m - par) / s
    f = s * (z * norm_cdf(z) + norm_pdf(z))
    if (f < 0).any():
        raise ValueError
    return f


def log_ei(m, v, eta, par=0.0):
    """robo/acquisition_functions/log_ei.py:74-120, branch for branch, calling
    scipy.stats.norm exactly like the reference (np.Infinity read as np.inf: it no
    longer exists in NumPy 2).  Bit-identical to the reference class on the same
    (m, v) -- tests/test_oracle.py."""
    from scipy.stats import norm
    f_min = eta - par
    s = np.sqrt(v)
    with np.errstate(divide="ignore", invalid="ignore"):
        z = (f_min - m) / s
    out = np.zeros([m.size])
    for i in range(m.size):
        mu, sigma = m[i], s[i]
        if abs(f_min - mu) == 0:
            out[i] = np.log(sigma) + norm.logpdf(z[i]) if sigma > 0 else -np.inf
        elif sigma == 0:
            out[i] = np.log(f_min - mu) if mu < f_min else -np.inf
        else:
            b = np.log(sigma) + norm.logpdf(z[i])
            if f_min > mu:
                a = np.log(f_min - mu) + norm.logcdf(z[i])
                out[i] = max(a, b) + np.log(1 + np.exp(-abs(b - a)))
            else:
                a = np.log(mu - f_min) + norm.logcdf(z[i])
                out[i] = -np.inf if a >= b else b + np.log(1 - np.exp(a - b))
    return out


def log_ei_vec(m, v, eta, par=0.0):
    """Vectorised form of :func:`log_ei` (same branches) on the erfc/erfcx formulas
    the device code uses (norm_logcdf above, within 5e-14 of scipy's).  For
    z < -30 the reference's ``b + log(1 - exp(a - b))`` cancels catastrophically
    (a, b ~ -z^2/2, a - b ~ 1/z^2), so one ulp of logcdf moves the result by ~1e-9
    relative there: compare with rtol 1e-8 in that tail, 1e-12 elsewhere."""
    f_min = eta - par
    s = np.sqrt(v)
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        z = (f_min - m) / s
        b = np.log(s) + norm_logpdf(z)
        lc = norm_logcdf(z)
        a_pos = np.log(f_min - m) + lc
        a_neg = np.log(m - f_min) + lc
        normal_pos = np.maximum(a_pos, b) + np.log(1 + np.exp(-np.abs(b - a_pos)))
        normal_neg = np.where(a_neg >= b, -np.inf, b + np.log(1 - np.exp(a_neg - b)))
        out = np.where(f_min > m, normal_pos, normal_neg)
        out = np.where(s == 0, np.where(m < f_min, np.log(f_min - m), -np.inf), out)
        out = np.where(f_min - m == 0, np.where(s > 0, b, -np.inf), out)
    return out


__all__.append("log_ei_vec")


def pi(m, v, eta, par=0.0):
    """robo/acquisition_functions/pi.py:57-63."""
    s = np.sqrt(v)
    return norm_cdf((eta - m - par) / s)


def lcb(m, v, par=1.0):
    """robo/acquisition_functions/lcb.py:62-65."""
    return -(m - par * np.sqrt(v))


def marginalize(acq_values):
    """robo/acquisition_functions/marginalization.py:115-121: (S,M) -> mean over S."""
    return np.asarray(acq_values).mean(axis=0)


def np_argmax(y):
    """RandomSampling.maximize's selection, robo/maximizers/random_sampling.py:48-50."""
    return int(np.argmax(y))

"""Expectation propagation for p_min = P(x_i is the minimiser) of a joint Gaussian belief over
Nb representer points, with the derivatives entropy search needs (Cunningham, Hennig &
Lacoste-Julien 2011, "Gaussian probabilities and expectation propagation"; Hennig & Schuler 2012).

Host code by design (Nb = 50, inherently sequential rank-1 site
updates, runs once per ``InformationGain.update``); same entry point and return convention as
``robo/util/epmgp.py:11-81`` -- ``joint_min(mu, var, with_derivatives)`` ->
``logP (N,)`` or ``(logP, dlogPdMu (N,N), dlogPdSigma (N, N(N+1)/2), dlogPdMudMu (N,N,N))`` with the
covariance derivative packed as the row-major LOWER triangle (what
``InformationGain._dh_fun`` contracts it with, information_gain.py:176).

Structure of the computation for candidate minimiser k: N-1 half-space factors
"f_l >= f_k", each approximated by a Gaussian site in the direction c_l = (e_l - e_k)/sqrt(2)
with precision p_l and shift mp_l; sites are refined by moment matching on the cavity until
the largest site change in a sweep is below 1e-3 (at most 50 sweeps); the normaliser logZ_k and
its derivatives w.r.t. the belief's mean and covariance follow in closed form from the final
sites.  Constants (1e-25 guard, float32-eps floor on site updates, |z| > 6 cut-offs, -500 floor
on log p) are the reference's, because they change the returned numbers.
"""
import numpy as np
from scipy import special

SQRT2 = np.sqrt(2.0)
EPS32 = np.finfo(np.float32).eps
LOG_2PI = np.log(2.0) + np.log(np.pi)


def _truncated_moment(z):
    """(phi/Phi ratio, log Phi(z), flag): flag -1 left of -6 (factor certainly violated), +1 right of
    +6 (factor inactive), 0 otherwise (robo/util/epmgp.py:240-249)."""
    if z < -6:
        return 1.0, -1.0e12, -1
    if z > 6:
        return 0.0, 0.0, 1
    log_pdf = -0.5 * (z * z + LOG_2PI)
    log_cdf = np.log(0.5 * special.erfc(-z / SQRT2))
    return np.exp(log_pdf - log_cdf), log_cdf, 0


class _Sites(object):
    """EP state for one candidate minimiser k."""

    def __init__(self, mu, Sigma, k):
        self.k = k
        self.n = mu.shape[0]
        self.M = mu.astype(np.float64).copy()
        self.V = Sigma.astype(np.float64).copy()
        self.prec = np.zeros(self.n - 1)      # site precisions p_l
        self.shift = np.zeros(self.n - 1)     # site precision-means mp_l
        self.log_scale = np.zeros(self.n - 1)
        self.failed = False

    def refine(self, idx, l):
        """moment-match the site of factor f_l >= f_k; returns the convergence measure d (NaN = dead)."""
        M, V, k = self.M, self.V, self.k
        p, mp = self.prec[idx], self.shift[idx]
        cVc = (V[l, l] - 2.0 * V[k, l] + V[k, k]) / 2.0
        Vc = (V[:, l] - V[:, k]) / SQRT2
        cM = (M[l] - M[k]) / SQRT2
        cav_var = max(cVc / (1.0 - p * cVc), 0.0)
        cav_mean = cM + cav_var * (p * cM - mp)
        with np.errstate(invalid="ignore", divide="ignore"):
            z = cav_mean / np.sqrt(cav_var + 1e-25)
        if np.isnan(z):
            z = -np.inf
        ratio, log_cdf, flag = _truncated_moment(z)
        if flag == -1:
            self.failed = True
            return np.nan
        if flag == 0:
            alpha = ratio / np.sqrt(cav_var)
            beta = alpha * (alpha * cav_var + cav_mean)
            r = beta / (1.0 - beta)
            p_new = r / cav_var
            mp_new = r * (alpha + cav_mean / cav_var) + alpha
            dp = max(-p + EPS32, p_new - p)
            dmp = max(-mp + EPS32, mp_new - mp)
            d = max(dmp, dp)
            p_out, mp_out = p + dp, mp + dmp
            log_s = log_cdf - 0.5 * (np.log(beta) - np.log(p_out) - np.log(cav_var)) \
                + (alpha * alpha) / (2.0 * beta) * cav_var
        else:   # factor inactive: remove its message
            dp, dmp = -p, -mp
            d = max(dmp, dp)
            p_out, mp_out, log_s = 0.0, 0.0, 0.0
        denom = 1.0 + dp * cVc
        V_new = V - dp / denom * np.outer(Vc, Vc)
        if np.any(np.isnan(V_new)):
            raise Exception("an error occurs while running expectation propagation in entropy search. "
                            "Resulting variance contains NaN")
        self.M = M + (dmp - cM * dp) / denom * Vc
        self.V = V_new
        self.prec[idx], self.shift[idx], self.log_scale[idx] = p_out, mp_out, log_s
        return d

    def run(self):
        others = [l for l in range(self.n) if l != self.k]
        for _ in range(50):
            total = 0.0
            for idx, l in enumerate(others):
                d = self.refine(idx, l)
                if np.isnan(d):
                    return
                total += abs(d)
            if abs(total) < 0.001:
                return


def _pack_lower(S):
    """row-major lower triangle (incl. diagonal) of a symmetric matrix -> (n(n+1)/2,)"""
    return S[np.tril_indices(S.shape[0])]


def _log_normaliser(mu, Sigma, k, with_derivatives):
    """log Z_k (and derivatives w.r.t. mu, mu mu, Sigma) of the product of the N-1 half-space sites."""
    n = mu.shape[0]
    st = _Sites(mu, Sigma, k)
    st.run()
    if st.failed:
        if not with_derivatives:
            return -np.inf
        return -np.inf, np.zeros(n), np.zeros((n, n)), np.zeros(n * (n + 1) // 2)
    C = np.eye(n) / SQRT2
    C[k, :] = -1.0 / SQRT2
    C = np.delete(C, k, 1)                       # (n, n-1): column per factor
    R = np.sqrt(st.prec)[None, :] * C
    r = np.sum(st.shift[None, :] * C, axis=1)
    nz = st.shift != 0
    mpm = np.sum(st.shift[nz] * st.shift[nz] / st.prec[nz])
    inner = np.eye(n - 1) + R.T.dot(Sigma).dot(R)
    rSr = r.dot(Sigma).dot(r)
    A = R.dot(np.linalg.solve(inner, R.T))
    A = 0.5 * (A.T + A)
    b = mu + Sigma.dot(r)
    Ab = A.dot(b)
    chol = None
    for jitter in (0.0, 1e-10, 1e-6):
        try:
            chol = np.linalg.cholesky(inner + jitter * np.eye(n - 1))
            break
        except np.linalg.LinAlgError:
            if jitter == 1e-6:
                raise
    log_det = 2.0 * np.sum(np.log(np.diagonal(chol)))
    logZ = 0.5 * (rSr - b.dot(Ab) - log_det) + mu.dot(r) + np.sum(st.log_scale) - 0.5 * mpm
    if not with_derivatives:
        return logZ
    d_mu = r - Ab
    d_mumu = -A
    btA = b.dot(A)
    dS = -A - 2.0 * np.outer(r, Ab) + np.outer(r, r) + np.outer(btA, Ab)
    dS = 0.5 * (dS + dS.T - np.diag(np.diagonal(dS)))
    return logZ, d_mu, d_mumu, _pack_lower(dS)


def joint_min(mu, var, with_derivatives=False, **kwargs):
    """log p_min over the N points of a Gaussian belief N(mu, var) (+ derivatives)."""
    mu = np.asarray(mu, dtype=np.float64)
    var = np.asarray(var, dtype=np.float64)
    n = mu.shape[0]
    logP = np.zeros(n)
    if with_derivatives:
        d_mu = np.zeros((n, n))
        d_sigma = np.zeros((n, n * (n + 1) // 2))
        d_mumu = np.zeros((n, n, n))
    for i in range(n):
        res = _log_normaliser(mu, var, i, with_derivatives)
        if with_derivatives:
            logP[i], d_mu[i], d_mumu[i], d_sigma[i] = res
        else:
            logP[i] = res
    logP[np.isinf(logP)] = -500
    raw = logP.copy()
    Z = np.sum(np.exp(raw))
    top = np.max(logP)
    s = top + np.log(np.sum(np.exp(logP - top)))
    s = top if np.isinf(s) else s
    logP = logP - s
    if not with_derivatives:
        return logP
    w = np.exp(raw) / Z                           # p_min itself
    Zm = w.dot(d_mu)                              # E_p[dlogZ/dmu]
    Zs = w.dot(d_sigma)
    outer = np.einsum('ki,kj->kij', d_mu, d_mu)
    gg = np.einsum('kij,k->ij', d_mumu + outer, w)
    # The second-derivative correction should be -gg + Zm Zm^T.  The reference evaluates
    # ``Zm.T * Zm`` on a 1-D array (robo/util/epmgp.py:73), i.e. the ELEMENT-WISE square broadcast
    # over rows, not the outer product.  Mirrored (it changes every information-gain value the
    # reference produces); DESIGN.md "Mirrored quirks".
    return logP, d_mu - Zm, d_sigma - Zs, d_mumu + (-gg + (Zm * Zm)[None, :])[None, :, :]

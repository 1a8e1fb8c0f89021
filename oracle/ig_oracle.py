"""NumPy restatement of the entropy-search information gain (TEST INFRASTRUCTURE ONLY).

Follows robo/acquisition_functions/information_gain.py: ``innovations`` (:253-272),
``_dh_fun`` (:169-203), ``loss_function`` (:68-72), ``update``'s W (:162-165) and the NaN/inf
guard of ``compute`` (:119-120).  The reference module itself cannot be imported here (it
imports emcee at module level, information_gain.py:5), so this half is "parity unpinned" at the
class level; the EP half it consumes IS pinned: robo_amd/util/epmgp.py is compared with the
reference's own robo/util/epmgp.py (importable) in tests/test_infogain.py.
"""
import sys

import numpy as np
import scipy.stats


def outcome_quantiles(Np):
    """information_gain.py:162-165"""
    return scipy.stats.norm.ppf(np.linspace(1. / (Np + 1), 1 - 1. / (Np + 1), Np))[np.newaxis, :]


def innovations(v, sigma_x_rep, sn2):
    """v: predictive variance at x (scalar), sigma_x_rep: (Nb, 1) cov(x, rep) -> (dM (Nb,1), dV (Nb,Nb))"""
    v = np.array([[v]], dtype=np.float64)
    v_ = v - sn2
    norm_cov = np.dot(sigma_x_rep, np.linalg.inv(v_))
    dm_rep = np.dot(norm_cov, np.linalg.cholesky(v + 1e-10))
    dv_rep = -norm_cov.dot(sigma_x_rep.T)
    return dm_rep, dv_rep


def dh_fun(v, sigma_x_rep, sn2, logP, lmb, dlogPdMu, dlogPdSigma, dlogPdMudMu, W):
    N = logP.size
    dMdx, dVdx = innovations(v, sigma_x_rep, sn2)
    dVdx = dVdx[np.triu(np.ones((N, N))).T.astype(bool), np.newaxis]
    dMM = dMdx.dot(dMdx.T)
    trterm = np.sum(np.sum(np.multiply(dlogPdMudMu, np.reshape(dMM, (1, dMM.shape[0], dMM.shape[1]))), 2), 1)[
        :, np.newaxis]
    logP = np.reshape(logP, (N, 1))
    lmb = np.reshape(lmb, (N, 1))
    detchange = dlogPdSigma.dot(dVdx) + 0.5 * trterm
    stochange = (dlogPdMu.dot(dMdx)).dot(W)
    lPred = np.add(logP + detchange, stochange)
    _max = np.amax(lPred, axis=0)
    with np.errstate(all="ignore"):
        s = _max + np.log(np.sum(np.exp(lPred - _max), axis=0))
    lsel = _max if np.any(np.isinf(s)) else s
    lPred = np.subtract(lPred, lsel)
    H = -np.sum(np.multiply(np.exp(logP), (logP + lmb)))
    dHp = -(-np.sum(np.multiply(np.exp(lPred), np.add(lPred, lmb)), axis=0) - H)
    dH = np.mean(dHp)
    if np.isnan(dH) or dH == np.inf:
        dH = -sys.float_info.max
    return dH


def innovation_inputs(ogp, X_test, zb, chunk=2048):
    """The two posterior quantities ``innovations`` (information_gain.py:253-272) reads for every candidate x,
    from an :class:`oracle.gp_oracle.OracleGP` -- the oracle's OWN numbers, nothing from the device:

      v[c]     = ``model.predict(x)[1]``                      (:255; diagonal, floored at eps)
      S[c, b]  = ``model.predict_variance(rep, x)[b, 0]``     (:263 -> gaussian_process.py:243-248: the last row
                 of the full covariance of [rep; x], floored at eps -- negative covariances included, :290-294)

    The reference forms one (Nb+1) x (Nb+1) covariance per candidate through ``gp.predict``
    (K** - K* K^-1 K*^T); the same entries are taken here from V = L^-1 K*^T, chunked over candidates:
    S = k(x, rep) - V_x^T V_rep.  Inputs in the caller's space (normalised like ``predict`` does)."""
    from oracle import gp_oracle as O
    import scipy.linalg as sla
    eps = float(np.finfo(np.float64).eps)
    norm = (lambda X: O.zero_one_normalization(X, ogp.lower, ogp.upper)[0]) if ogp.normalize_input else (lambda X: X)
    scale = ogp.y_std ** 2 if ogp.normalize_output else 1.0
    th_k = ogp.theta[:-1]
    Zn = norm(np.asarray(zb, dtype=np.float64))
    Vz = sla.solve_triangular(ogp.L, O.kernel_matrix(ogp.kind, th_k, ogp.X, Zn), lower=True, check_finite=False)
    M = X_test.shape[0]
    v = np.empty(M)
    S = np.empty((M, Zn.shape[0]))
    for s in range(0, M, chunk):
        Xn = norm(np.asarray(X_test[s:s + chunk], dtype=np.float64))
        Vx = sla.solve_triangular(ogp.L, O.kernel_matrix(ogp.kind, th_k, ogp.X, Xn), lower=True, check_finite=False)
        v[s:s + chunk] = (O.kernel_diag(ogp.kind, th_k, Xn) - np.sum(Vx * Vx, axis=0)) * scale
        S[s:s + chunk] = (O.kernel_matrix(ogp.kind, th_k, Xn, Zn) - Vx.T @ Vz) * scale
    return np.clip(v, eps, np.inf), np.clip(S, eps, np.inf)

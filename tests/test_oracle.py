"""Pin the oracle (oracle/gp_oracle.py) before anything is compared against it.

* acquisition half: against fixtures produced by the REFERENCE'S OWN classes
  (tests/golden/make_golden.py) -- bit-level/1e-13 agreement required;
* GP half: the reference's only numeric pin on the george boundary, the posterior
  identity of test/test_models/test_gaussian_process.py:44-49, plus the LCB value
  pin (test_lcb.py:25) and incumbent == argmin (test_gaussian_process.py:77-82).
"""
import os

import numpy as np
import pytest
from scipy.stats import norm

from oracle import gp_oracle as O
from _tol import assert_logei_close
from make_golden import CASES, golden_inputs, mcmc_inputs


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"))


def test_norm_formulas_match_scipy():
    z = np.concatenate([np.linspace(-38, 9, 20001), [0.0, -1.0, 1.0]])
    assert np.allclose(O.norm_cdf(z), norm.cdf(z), rtol=1e-13, atol=0)
    assert np.allclose(O.norm_pdf(z), norm.pdf(z), rtol=1e-13, atol=0)
    assert np.allclose(O.norm_logpdf(z), norm.logpdf(z), rtol=1e-14, atol=0)
    lc = norm.logcdf(z)
    assert np.all(np.abs(O.norm_logcdf(z) - lc) <= 5e-14 * np.abs(lc) + 1e-300)


def test_demo_model_pins(golden_dir):
    """Values quoted in SURVEY.md 8(c): 0.0204 / -3.891 / 0.1217 / 0.0168."""
    g = _load(golden_dir, "demo_model_pins")
    y = g["y"]
    m = np.full(5, y.mean())
    v = np.full(5, y.var())
    eta = y.min()
    assert abs(g["ei"][0] - 0.0204) < 1e-4 and abs(g["log_ei"][0] + 3.891) < 1e-3
    np.testing.assert_allclose(O.ei(m, v, eta), g["ei"], rtol=1e-13)
    np.testing.assert_allclose(O.log_ei(m, v, eta), g["log_ei"], rtol=1e-13)
    np.testing.assert_allclose(O.pi(m, v, eta), g["pi"], rtol=1e-13)
    np.testing.assert_allclose(O.lcb(m, v), g["lcb"], rtol=1e-13)
    # reference pin test/test_acquisition_functions/test_lcb.py:25
    np.testing.assert_almost_equal(g["lcb"][0], -np.mean(y) + np.std(y), decimal=3)


def test_acq_elementwise_against_reference(golden_dir):
    g = _load(golden_dir, "acq_elementwise")
    m, v, eta = g["m"], g["v"], float(g["eta"])
    with np.errstate(all="ignore"):
        np.testing.assert_array_equal(O.log_ei(m, v, eta), g["log_ei"])
        np.testing.assert_array_equal(O.log_ei(m, v, eta, par=0.1), g["log_ei_par"])
        assert_logei_close(O.log_ei_vec(m, v, eta), g["log_ei"], (eta - m) / np.sqrt(v))
        np.testing.assert_allclose(O.pi(m, v, eta), g["pi"], rtol=1e-13, atol=0)
        np.testing.assert_allclose(O.lcb(m, v), g["lcb"], rtol=1e-15)
    pos = g["pos"]
    np.testing.assert_allclose(O.ei(m[pos], v[pos], eta), g["ei_pos"], rtol=1e-12, atol=1e-300)
    # ei.py:72-74: one zero-sigma point collapses the whole batch
    assert O.ei(m, v, eta).shape == (1, 1) and g["ei_collapsed"].shape == (1, 1)
    assert np.isneginf(g["log_ei"]).any()


@pytest.mark.parametrize("name", list(CASES))
def test_gp_cases_regression(golden_dir, name):
    g = _load(golden_dir, name)
    inp = golden_inputs(name)
    gp = O.OracleGP(inp["kind"], inp["theta"], normalize_output=inp["nout"],
                    lower=inp["lower"], upper=inp["upper"])
    gp.train(inp["X"], inp["y"])
    mu, var = gp.predict(inp["Xc"], diag_only=True)
    np.testing.assert_allclose(mu, g["mu"], rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(var, g["var"], rtol=1e-9, atol=1e-13)
    _, eta = gp.get_incumbent()
    assert eta == g["eta"]
    # oracle acquisitions on the stored (mu, var) == the reference classes' output
    np.testing.assert_allclose(O.ei(g["mu"], g["var"], eta), g["ei"], rtol=1e-12, atol=1e-300)
    assert_logei_close(O.log_ei_vec(g["mu"], g["var"], eta), g["log_ei"],
                       (eta - g["mu"]) / np.sqrt(g["var"]))
    np.testing.assert_allclose(O.pi(g["mu"], g["var"], eta), g["pi"], rtol=1e-12, atol=1e-300)
    np.testing.assert_allclose(O.lcb(g["mu"], g["var"]), g["lcb"], rtol=1e-14)
    np.testing.assert_allclose(O.ei(g["mu"], g["var"], eta, par=0.3), g["ei_par"], rtol=1e-12,
                               atol=1e-300)
    assert O.np_argmax(O.ei(g["mu"], g["var"], eta)) == int(g["argmax_ei"])
    if "var_fullcov_path" in g:
        # diag-only path == the reference's full-covariance-then-diag call sequence
        np.testing.assert_allclose(g["var"], g["var_fullcov_path"], rtol=1e-7, atol=1e-10)
        np.testing.assert_allclose(g["mu"], g["mu_fullcov_path"], rtol=1e-12, atol=1e-13)


def test_posterior_identity_reference_pin():
    """test/test_models/test_gaussian_process.py:44-49 restated: the oracle's
    predictive covariance equals K_zz - K_zx (K_xx + noise I)^-1 K_zx^T (MSE < 1e-4)."""
    rs = np.random.RandomState(3)
    X = rs.rand(10, 2)
    y = np.sinc(X * 10 - 5).sum(axis=1)
    theta = np.array([np.log(2.0 / 2), 0.0, 0.0, np.log(1e-3)])
    gp = O.OracleGP("matern52", theta, lower=np.zeros(2), upper=np.ones(2))
    gp.train(X, y)
    Xt = rs.rand(10, 2)
    _, v = gp.predict(Xt, full_cov=True)
    K_zz = O.kernel_matrix("matern52", theta[:-1], Xt)
    K_zx = O.kernel_matrix("matern52", theta[:-1], Xt, X)
    K_nz = O.kernel_matrix("matern52", theta[:-1], X) + (gp.noise + O.JITTER) * np.eye(10)
    var = K_zz - K_zx @ np.linalg.inv(K_nz) @ K_zx.T
    assert np.mean((np.clip(var, O.EPS, np.inf) - v) ** 2) < 1e-4
    # test_gaussian_process.py:77-82
    inc, inc_val = gp.get_incumbent()
    b = np.argmin(y)
    np.testing.assert_almost_equal(inc, X[b], decimal=5)
    assert inc_val == y[b]


def test_kernel_contract():
    """metric = squared length scale; k(x,x) = amp; Matern-5/2 closed form."""
    th = np.array([np.log(1.7), np.log(0.3), np.log(2.0)])
    x1 = np.array([[0.1, 0.2]])
    x2 = np.array([[0.4, -0.5]])
    r2 = (0.3 ** 2) / 0.3 + (0.7 ** 2) / 2.0
    k = O.kernel_matrix("matern52", th, x1, x2)[0, 0]
    assert abs(k - 1.7 * (1 + np.sqrt(5 * r2) + 5 * r2 / 3) * np.exp(-np.sqrt(5 * r2))) < 1e-15
    assert abs(O.kernel_matrix("rbf", th, x1, x2)[0, 0] - 1.7 * np.exp(-r2 / 2)) < 1e-15
    assert O.kernel_matrix("matern52", th, x1)[0, 0] == pytest.approx(1.7, abs=1e-15)


def test_loglik_against_dense_formula():
    inp = golden_inputs("small_matern")
    gp = O.OracleGP(inp["kind"], inp["theta"], lower=inp["lower"], upper=inp["upper"])
    gp.train(inp["X"], inp["y"])
    K = O.kernel_matrix(inp["kind"], inp["theta"][:-1], gp.X) + (gp.noise + O.JITTER) * np.eye(40)
    r = gp.y - gp.mean
    ll = -0.5 * (r @ np.linalg.solve(K, r) + np.linalg.slogdet(K)[1] + 40 * np.log(2 * np.pi))
    assert abs(gp.loglikelihood(inp["theta"]) - ll) < 1e-9
    assert gp.nll(inp["theta"]) == -gp.loglikelihood(inp["theta"])
    assert gp.nll(np.full(5, 21.0)) == 1e25 and gp.loglikelihood(np.full(5, -21.0)) == -np.inf


def test_mcmc_marginal_regression(golden_dir):
    g = _load(golden_dir, "mcmc_marginal")
    inp = mcmc_inputs()
    eta = inp["y"].min()
    acq = np.array([O.log_ei_vec(g["mu_s"][s], g["var_s"][s], eta) for s in range(6)])
    np.testing.assert_allclose(O.marginalize(acq), g["marg_log_ei"], rtol=1e-8)
    acq = np.array([O.ei(g["mu_s"][s], g["var_s"][s], eta) for s in range(6)])
    np.testing.assert_allclose(O.marginalize(acq), g["marg_ei"], rtol=1e-12, atol=1e-300)
    m, v = O.mcmc_mixture(g["mu_s"], g["var_s"])
    np.testing.assert_array_equal(m, g["mix_m"])
    np.testing.assert_array_equal(v, g["mix_v"])


@pytest.mark.parametrize("kind", ["matern52", "rbf"])
def test_oracle_agrees_with_scikit_learn(kind):
    """Independent third-party cross-check of everything the oracle DEFINES rather than restates
    (george is absent, so its kernel values are 'parity unpinned'): scikit-learn's Matern(nu=2.5) / RBF
    with length_scale = sqrt(metric), ConstantKernel(amp), WhiteKernel(sigma^2 + jitter) and its
    GaussianProcessRegressor give the same kernel matrix, posterior mean / variance, log marginal
    likelihood and likelihood gradient (d/d log ell = 2 d/d log m; the noise entry differs by the factor
    sigma^2 the reference's grad_nll drops, gaussian_process.py:178-182)."""
    sk = pytest.importorskip("sklearn.gaussian_process")
    from sklearn.gaussian_process.kernels import RBF, ConstantKernel, Matern, WhiteKernel
    rs = np.random.RandomState(0)
    N, D = 60, 3
    X = rs.rand(N, D)
    y = np.sin(3 * X.sum(axis=1))
    theta = np.concatenate([[0.3], np.log([0.2, 0.5, 0.9]), [np.log(1e-2)]])
    amp, m, s2 = np.exp(theta[0]), np.exp(theta[1:-1]), np.exp(theta[-1]) + O.JITTER
    base = Matern(length_scale=np.sqrt(m), nu=2.5) if kind == "matern52" else RBF(length_scale=np.sqrt(m))
    k = ConstantKernel(amp) * base
    np.testing.assert_allclose(O.kernel_matrix(kind, theta[:-1], X), k(X), rtol=0, atol=1e-14)
    gpr = sk.GaussianProcessRegressor(kernel=k + WhiteKernel(s2), alpha=0.0, optimizer=None)
    c = y.mean()
    gpr.fit(X, y - c)
    Xs = rs.rand(9, D)
    mu, std = gpr.predict(Xs, return_std=True)
    L = O.gp_compute(kind, theta, X)
    mu_o, var_o = O.gp_predict_diag(kind, theta, L, X, y, c, Xs)
    np.testing.assert_allclose(mu_o, mu + c, rtol=0, atol=1e-12)
    np.testing.assert_allclose(var_o, std ** 2 - s2, rtol=0, atol=1e-12)   # sklearn adds the white noise
    lml, grad = gpr.log_marginal_likelihood(gpr.kernel_.theta, eval_gradient=True)
    np.testing.assert_allclose(O.gp_log_likelihood(L, y, c), lml, rtol=1e-13)
    g = O.gp_grad_log_likelihood(kind, theta, X, y, c)
    np.testing.assert_allclose(g[0], grad[0], rtol=1e-9)
    np.testing.assert_allclose(2.0 * g[1:-1], grad[1:-1], rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(g[-1] * s2, grad[-1], rtol=1e-9)


def test_kernel_gradient_matches_central_differences():
    """kernel_gradient (george's kernel.gradient, gaussian_process.py:181) is the exact derivative of
    kernel_matrix for every kind, Fabolas product kernel included"""
    rs = np.random.RandomState(1)
    for kind, D in (("matern52", 3), ("rbf", 2), ("fabolas", 4)):
        X = rs.rand(25, D)
        P = O.n_kernel_params(kind, D)
        th = 0.3 * rs.randn(P)
        G = O.kernel_gradient(kind, th, X)
        for p in range(P):
            e = np.zeros(P)
            e[p] = 1e-6
            fd = (O.kernel_matrix(kind, th + e, X) - O.kernel_matrix(kind, th - e, X)) / 2e-6
            np.testing.assert_allclose(G[:, :, p], fd, rtol=0, atol=2e-8 * max(1.0, np.abs(fd).max()))


# ------------------------------------------------------------------------------------------------
# GP half against the REFERENCE'S OWN classes (tests/golden/make_golden_ref.py: robo.models.* and
# robo.acquisition_functions.information_gain* executed unchanged on the george/emcee stand-ins)
# ------------------------------------------------------------------------------------------------
def test_oracle_gp_matches_reference_class_fixtures(golden_dir):
    """OracleGP (the restatement used as checker at sizes without a fixture and as cpu_baseline) against
    what robo.models.gaussian_process.GaussianProcess itself returned: train/predict/nll/grad_nll/incumbent"""
    import make_golden_ref as G
    for name in G.GP_CASES:
        inp, gold = G.ref_inputs(name), _load(golden_dir, name)
        gp = O.OracleGP(inp["kind"], inp["theta"], normalize_output=inp["nout"], lower=inp["lower"],
                        upper=inp["upper"])
        gp.train(inp["X"], inp["y"])
        mu, var = gp.predict(inp["Xc"], diag_only=True)
        np.testing.assert_allclose(mu, gold["mu"], rtol=1e-10, atol=1e-11)
        np.testing.assert_allclose(var, gold["var"], rtol=0, atol=1e-10)
        inc, val = gp.get_incumbent()
        np.testing.assert_array_equal(inc, gold["inc"])
        assert val == gold["inc_val"]
        _, cov = gp.predict(inp["Xc"][:33], full_cov=True)
        np.testing.assert_allclose(cov, gold["cov33"], rtol=0, atol=1e-10)
        if "nll" in gold.files:
            np.testing.assert_allclose([gp.nll(t) for t in gold["nll_thetas"]], gold["nll"], rtol=1e-12)
            for t, ref in zip(gold["nll_thetas"][:3], gold["grad_nll"]):
                g = -O.gp_grad_log_likelihood(inp["kind"], t, gp.X, gp.y, gp.mean)
                np.testing.assert_allclose(g, ref, rtol=1e-9, atol=1e-10 * np.abs(ref).max())


def test_refstub_kernels_equal_oracle_kernels():
    """two independently written evaluations of the kernel contract: generic composition in
    oracle/refstub/george/kernels.py vs the (kind, theta) form in oracle/gp_oracle.py, values and gradients"""
    import sys
    stub = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "refstub")
    sys.path.insert(0, stub)
    try:
        import george
    finally:
        sys.path.remove(stub)
    rs = np.random.RandomState(0)
    D = 4
    X1, X2 = rs.rand(23, D), rs.rand(17, D)
    th = np.concatenate([[0.4], 0.5 * rs.randn(D)])
    for kind, cls in (("matern52", george.kernels.Matern52Kernel), ("rbf", george.kernels.ExpSquaredKernel)):
        k = 2.0 * cls(np.ones(D), ndim=D)
        assert len(k) == D + 1 and np.isclose(k.get_parameter_vector()[0], np.log(2.0 / D))
        k.set_parameter_vector(th)
        np.testing.assert_allclose(k.get_value(X1, X2), O.kernel_matrix(kind, th, X1, X2), rtol=1e-13)
        np.testing.assert_allclose(k.gradient(X1), O.kernel_gradient(kind, th, X1), rtol=1e-11, atol=1e-14)
    # Fabolas product, built the way robo/fmin/fabolas.py:103-117 builds it
    kern = 1
    for d in range(D - 1):
        kern *= george.kernels.Matern52Kernel(np.ones([1]) * 0.01, ndim=D, axes=d)
    kern *= george.kernels.BayesianLinearRegressionKernel(log_a=0.1, log_b=0.1, ndim=D, axes=D - 1)
    assert len(kern) == 1 + (D - 1) + 2
    thf = np.concatenate([[-0.3], 0.5 * rs.randn(D - 1), [0.2, -0.4]])
    kern.set_parameter_vector(thf)
    np.testing.assert_allclose(kern.get_value(X1, X2), O.kernel_matrix("fabolas", thf, X1, X2), rtol=1e-13)
    np.testing.assert_allclose(kern.gradient(X1), O.kernel_gradient("fabolas", thf, X1), rtol=1e-11, atol=1e-14)


def test_ig_oracle_matches_reference_class_fixture(golden_dir):
    """oracle/ig_oracle.py (restatement of _dh_fun / innovations) against InformationGain.compute itself"""
    import make_golden_ref as G
    from oracle import ig_oracle as IG
    inp, gold = G.infogain_inputs(), _load(golden_dir, "ref_infogain")
    gp = O.OracleGP("matern52", inp["theta"], lower=inp["lower"], upper=inp["upper"])
    gp.train(inp["X"], inp["y"])
    W = IG.outcome_quantiles(400)
    zb = gold["zb"]
    out = np.empty(60)
    for c in range(60):
        x = inp["Xc"][c:c + 1]
        _, v = gp.predict(x)
        _, cov = gp.predict(np.concatenate((zb, x)), full_cov=True)
        out[c] = IG.dh_fun(v[0], cov[-1, :-1, None], float(gold["sn2"]), gold["logP"], gold["lmb"], gold["dlogPdMu"],
                           gold["dlogPdSigma"], gold["dlogPdMudMu"], W)
    np.testing.assert_allclose(out, gold["ig"][:60], rtol=1e-9, atol=1e-12)


def test_oracle_predictive_gradients_match_central_differences():
    """the checker of robo_gp_predict_grad: analytic input gradients of the oracle posterior == central
    differences of the oracle's own predict, every kernel kind, input + output normalisation"""
    rs = np.random.RandomState(3)
    for kind, D, nout in (("matern52", 3, False), ("rbf", 4, True), ("fabolas", 4, False)):
        N = 40
        lower, upper = np.full(D, -1.0), np.full(D, 2.0)
        X = lower + (upper - lower) * rs.rand(N, D)
        y = np.sin(X.sum(axis=1)) * 2 + 0.3
        P = O.n_kernel_params(kind, D) + 1
        theta = 0.3 * rs.randn(P)
        theta[-1] = np.log(1e-2)
        norm_in = kind != "fabolas"
        gp = O.OracleGP(kind, theta, normalize_output=nout, normalize_input=norm_in, lower=lower, upper=upper)
        gp.train(X if norm_in else (X - lower) / (upper - lower), y)
        Xt = (lower + (upper - lower) * rs.rand(6, D)) if norm_in else rs.rand(6, D)
        dm, dv = gp.predictive_gradients(Xt)
        assert dm.shape == (6, D, 1) and dv.shape == (6, D)
        h = 1e-6
        for d in range(D):
            e = np.zeros(D)
            e[d] = h
            mp, vp = gp.predict(Xt + e, diag_only=True)
            mm, vm = gp.predict(Xt - e, diag_only=True)
            np.testing.assert_allclose(dm[:, d, 0], (mp - mm) / (2 * h), rtol=1e-6, atol=1e-7)
            np.testing.assert_allclose(dv[:, d], (vp - vm) / (2 * h), rtol=1e-5, atol=1e-7)

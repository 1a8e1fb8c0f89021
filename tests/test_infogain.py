"""Entropy search: EP vs the reference's own epmgp (importable), device information gain vs the
NumPy restatement (oracle/ig_oracle.py), through the interpreter on CPU and on the GPU."""
import os
import sys

import numpy as np
import pytest

from oracle import gp_oracle as O
from oracle import ig_oracle as IG
from robo_amd import _lib
from robo_amd.util import epmgp

HERE = os.path.dirname(os.path.abspath(__file__))
HAVE_REF = os.path.isdir("/root/reference/robo")


@pytest.mark.skipif(not HAVE_REF, reason="reference tree not on this box")
def test_ep_matches_reference_epmgp():
    if not hasattr(np, "Infinity"):
        np.Infinity = np.inf
    if not hasattr(np, "NAN"):
        np.NAN = np.nan
    sys.path.insert(0, "/root/reference")
    from robo.util import epmgp as ref
    rs = np.random.RandomState(0)
    for n in (3, 8, 20):
        A = rs.randn(n, n)
        S = A @ A.T / n + 0.1 * np.eye(n)
        mu = rs.randn(n)
        r = ref.joint_min(mu, S, with_derivatives=True)
        m = epmgp.joint_min(mu, S, with_derivatives=True)
        for a, b in zip(r, m):
            np.testing.assert_allclose(b, a, rtol=1e-10, atol=1e-13)
        np.testing.assert_allclose(epmgp.joint_min(mu, S), ref.joint_min(mu, S), rtol=1e-10, atol=1e-13)


def test_pmin_pins():
    """test/test_acquisition_functions/test_information_gain.py:33-56: uniform and Dirac beliefs"""
    n = 10
    p = np.exp(epmgp.joint_min(np.zeros(n), np.eye(n)))
    assert np.all(np.abs(p - 1.0 / n) < 0.03)
    mu = np.ones(n) * 1e4
    mu[0] = -1e4
    assert np.exp(epmgp.joint_min(mu, np.eye(n) * 1e-3))[0] == 1.0


def test_pmin_monte_carlo():
    """robo/util/mc_part.py joint_pmin: the reference's known answer (test/test_util/test_mc_part.py:9-14: two independent
    unit Gaussians -> [0.5, 0.5] to 10 % with 10 000 samples), agreement with the EP approximation on a correlated belief,
    the jitter ladder on a singular covariance, and -- where the reference tree is present -- its own numbers, bit for bit"""
    from robo_amd.util.mc_part import joint_pmin
    np.random.seed(1)
    np.testing.assert_allclose(joint_pmin(np.zeros([2, 1]), np.diag(np.ones(2)), 10000), [0.5, 0.5], rtol=1e-1)
    rs = np.random.RandomState(0)
    A = rs.randn(6, 6)
    V, m = A @ A.T / 6 + 0.05 * np.eye(6), rs.randn(6, 1)
    np.random.seed(2)
    mc = joint_pmin(m, V, 40000)
    assert abs(mc.sum() - 1.0) < 1e-12
    np.testing.assert_allclose(mc, np.exp(epmgp.joint_min(m[:, 0], V)), atol=0.03)
    singular = np.ones((4, 4))
    np.random.seed(4)
    p = joint_pmin(np.zeros((4, 1)), singular, 500)                # rank one: factorable only with the diagonal raised
    assert abs(p.sum() - 1.0) < 1e-12 and np.all(p > 0.15)
    dirac = joint_pmin(np.array([[-1e4], [1e4]]), np.eye(2) * 1e-3, 100)
    np.testing.assert_array_equal(dirac, [1.0, 1e-70])
    if HAVE_REF:
        sys.path.insert(0, "/root/reference")
        from robo.util.mc_part import joint_pmin as ref
        for seed, (mm, VV) in enumerate(((m, V), (np.zeros((4, 1)), singular))):
            np.random.seed(seed)
            want = ref(mm, VV, 1500)
            np.random.seed(seed)
            np.testing.assert_array_equal(joint_pmin(mm, VV, 1500), want)


def _setup(ctx, N=60, D=3, M=150, Nb=12, Np=40, seed=0, kind="matern52"):
    rs = np.random.RandomState(seed)
    X = rs.rand(N, D)
    y = np.sin(3 * X.sum(axis=1)) + 0.1 * rs.randn(N)
    if kind == "fabolas":   # BASELINE config 4: D-1 inputs + the basis-transformed fidelity column
        X[:, -1] = (1.0 - X[:, -1]) ** 2
        theta = np.concatenate([[0.0], np.log(0.3 * (D - 1)) + 0.2 * rs.randn(D - 1), [-0.5, 0.3], [np.log(1e-2)]])
    else:
        theta = np.concatenate([[0.0], np.log([0.3, 0.5, 0.8])[:D], [np.log(1e-2)]])
    Xc = rs.rand(M, D)
    zb = rs.rand(Nb, D)
    if kind == "fabolas":   # representers live on the s = 1 subspace (information_gain_per_unit_cost.py:125-138)
        zb[:, -1] = 0.0
    lmb = rs.randn(Nb)
    ogp = O.OracleGP(kind, theta, normalize_input=False)
    ogp.train(X, y)
    mu_b, var_b = ogp.predict(zb, full_cov=True)
    logP, dMu, dSig, dMM = epmgp.joint_min(mu_b, var_b, with_derivatives=True)
    W = IG.outcome_quantiles(Np)
    g = _lib.DeviceGP(ctx, kind, N, D)
    g.set_data(X, y)
    g.fit(theta, ogp.mean)
    return dict(ogp=ogp, g=g, Xc=Xc, zb=zb, lmb=lmb, logP=logP, dMu=dMu, dSig=dSig, dMM=dMM, W=W,
                sn2=np.exp(theta[-1]), ep=_lib.EPState(logP, lmb, W, dMu, dSig, dMM))


def _check_ig(ctx, **kw):
    d = _setup(ctx, **kw)
    cand = _lib.Candidates(ctx, d["Xc"])
    rep = _lib.Candidates(ctx, d["zb"])
    # (1) cross-covariances and variances of EVERY candidate vs the oracle's own (clipped at eps like the reference)
    S = _lib.cross_cov(d["g"], cand, rep)
    _, var = d["g"].predict(cand)
    var_o, S_o = IG.innovation_inputs(d["ogp"], d["Xc"], d["zb"])
    amp = float(np.max(O.kernel_diag(d["ogp"].kind, d["ogp"].theta[:-1], d["Xc"][:8])))
    np.testing.assert_allclose(S, S_o, rtol=0, atol=1e-9 * amp)
    np.testing.assert_allclose(var, var_o, rtol=0, atol=1e-9 * amp)
    # (2) information gain of every candidate vs the NumPy restatement fed with the ORACLE's (v, s): nothing the
    # device computed enters the expected values
    vals, mx, am = _lib.ig_eval(d["g"], cand, rep, d["ep"], d["sn2"])
    ref = np.array([IG.dh_fun(var_o[c], S_o[c][:, None], d["sn2"], d["logP"], d["lmb"], d["dMu"], d["dSig"], d["dMM"],
                              d["W"]) for c in range(d["Xc"].shape[0])])
    np.testing.assert_allclose(vals, ref, rtol=1e-6, atol=1e-6 * np.abs(ref).max())
    want = int(np.argmax(ref))
    srt = np.sort(ref)
    assert am == int(np.argmax(vals)) and (am == want or srt[-1] - srt[-2] <= 1e-6 * abs(ref[want])), (am, want)
    assert mx == vals[am]
    # (2b) the entropy algebra alone (same restatement on the device's own (v, s)): tighter, not a parity claim
    alg = np.array([IG.dh_fun(var[c], S[c][:, None], d["sn2"], d["logP"], d["lmb"], d["dMu"], d["dSig"], d["dMM"],
                              d["W"]) for c in range(0, d["Xc"].shape[0], 7)])
    np.testing.assert_allclose(vals[::7], alg, rtol=1e-8, atol=1e-10)
    # (3) the moments entry point (any model) gives the same numbers
    vals2 = _lib.ig_from_moments(ctx, S, var, d["ep"], d["sn2"])
    np.testing.assert_allclose(vals2, vals, rtol=1e-12, atol=1e-14)
    # (4) the representer points' solve is kept across calls on the same factor and redone after a refit or new points
    vals3, _, _ = _lib.ig_eval(d["g"], cand, rep, d["ep"], d["sn2"])          # served from the kept solve
    np.testing.assert_array_equal(vals3, vals)
    theta2 = d["ogp"].theta.copy()
    theta2[1] += 0.3
    d["g"].fit(theta2, d["ogp"].mean)
    vals4, _, _ = _lib.ig_eval(d["g"], cand, rep, d["ep"], d["sn2"])          # new factor: must not reuse
    rep_fresh = _lib.Candidates(ctx, d["zb"])
    vals5, _, _ = _lib.ig_eval(d["g"], cand, rep_fresh, d["ep"], d["sn2"])
    np.testing.assert_array_equal(vals4, vals5)
    assert np.max(np.abs(vals4 - vals)) > 0
    rep.set_points(d["zb"][::-1].copy())                                     # new points in the same handle
    rep_rev = _lib.Candidates(ctx, d["zb"][::-1].copy())
    vals6, _, _ = _lib.ig_eval(d["g"], cand, rep, d["ep"], d["sn2"])
    vals7, _, _ = _lib.ig_eval(d["g"], cand, rep_rev, d["ep"], d["sn2"])
    np.testing.assert_array_equal(vals6, vals7)
    # (5) the EP state's device copy is kept per candidate handle and re-made when ANY of its arrays changes
    dMM2 = d["dMM"] * 1.5                       # (same shapes, same small arrays: only the largest tensor differs)
    ep2 = _lib.EPState(d["logP"], d["lmb"], d["W"], d["dMu"], d["dSig"], dMM2)
    d["g"].fit(d["ogp"].theta, d["ogp"].mean)
    rep.set_points(d["zb"])
    v_a, _, _ = _lib.ig_eval(d["g"], cand, rep, d["ep"], d["sn2"])
    v_b, _, _ = _lib.ig_eval(d["g"], cand, rep, ep2, d["sn2"])
    v_c, _, _ = _lib.ig_eval(d["g"], cand, rep, d["ep"], d["sn2"])
    cand_fresh = _lib.Candidates(ctx, d["Xc"])
    v_d, _, _ = _lib.ig_eval(d["g"], cand_fresh, rep, ep2, d["sn2"])
    np.testing.assert_array_equal(v_a, vals)
    np.testing.assert_array_equal(v_c, vals)
    np.testing.assert_array_equal(v_b, v_d)
    assert np.max(np.abs(v_b - v_a)) > 0
    cand_fresh.close()
    rep_fresh.close()
    rep_rev.close()
    cand.close()
    rep.close()
    d["g"].close()


@pytest.fixture(scope="module")
def emu_ctx():
    sys.path.insert(0, os.path.join(HERE, "hipemu"))
    import build_emu
    _lib.use_library(build_emu.build())
    ctx = _lib.Context(0)
    yield ctx
    ctx.close()
    _lib.use_library(None)


def test_information_gain_logic_emulated(emu_ctx):
    _check_ig(emu_ctx, N=60, D=3, M=150, Nb=12, Np=40)
    _check_ig(emu_ctx, N=70, D=4, M=130, Nb=10, Np=30, seed=4, kind="fabolas")


def test_information_gain_class_emulated(emu_ctx):
    """the assertions of the reference's test_information_gain.py:21-31,58-72 on the RoBO-surface class"""
    from robo_amd.acquisition_functions import InformationGain
    from robo_amd.kernels import Matern52Kernel
    from robo_amd.models import GaussianProcess
    rs = np.random.RandomState(2)
    lo, hi = np.zeros(1), np.ones(1)
    X = rs.rand(10, 1)
    y = np.sinc(X * 10 - 5).sum(axis=1)
    model = GaussianProcess(2 * Matern52Kernel(np.array([0.1]), ndim=1), noise=1e-3, lower=lo, upper=hi,
                            rng=np.random.RandomState(3))
    model.train(X, y, do_optimize=False)
    a = InformationGain(model, lo, hi, Nb=10, Np=30, rng=np.random.RandomState(4))
    a.update(model)
    assert a.zb.shape == (10, 1) and np.all(a.zb >= lo) and np.all(a.zb <= hi)
    Xt = rs.rand(5, 1)
    v = a.compute(Xt)
    assert v.shape == (5,) and np.all(np.isfinite(v))
    out = a.compute(np.array([[1.5]]))
    assert out[0] == np.spacing(1)
    assert a.argmax(Xt) == int(np.argmax(v))
    # the reference's per-candidate public methods (information_gain.py:60-66, :199-272): dh_fun of one point is that
    # point's value in the batch; innovations + loss_function, put together the way the reference's entropy step does
    # on the host, give the same number as the fused device path
    for i in range(3):
        x = Xt[i:i + 1]
        np.testing.assert_array_equal(a.dh_fun(x), v[i:i + 1])
        dm, dv = a.innovations(x, a.zb)
        Nb = a.logP.size
        assert dm.shape == (Nb, 1) and dv.shape == (Nb, Nb)
        upper_by_column = dv[np.triu(np.ones((Nb, Nb))).T.astype(bool), np.newaxis]
        curvature = 0.5 * np.einsum("kij,ij->k", a.dlogPdMudMu, dm @ dm.T)[:, None]
        pred = a.logP + a.dlogPdSigma @ upper_by_column + curvature + (a.dlogPdMu @ dm) @ a.W
        top = pred.max(axis=0)
        pred = pred - (top + np.log(np.exp(pred - top).sum(axis=0)))
        gain = float(np.mean(-a.loss_function(a.logP, a.lmb, pred, a.zb)))
        np.testing.assert_allclose(gain, v[i], rtol=1e-6, atol=1e-12)
    far = a.dh_fun(np.array([[1.5]]))
    assert isinstance(far, tuple) and far[0][0, 0] == np.spacing(1)      # (the reference returns a pair outside the box)


@pytest.mark.gpu
def test_information_gain_gpu():
    _lib.use_library(None)
    ctx = _lib.Context(0)
    _check_ig(ctx, N=60, D=3, M=150, Nb=12, Np=40)
    _check_ig(ctx, N=700, D=3, M=3000, Nb=50, Np=400, seed=3)
    ctx.close()


@pytest.mark.gpu
def test_information_gain_config4_shard():
    """BASELINE config 4 at its per-GPU size: Fabolas product kernel, N = 4096, D = 10 + 1, Nb = 50,
    Np = 400, 65 536 / 8 = 8192 candidates -- cross-covariances against the oracle, information gain of
    every candidate against the NumPy restatement, argmax index equal."""
    _lib.use_library(None)
    ctx = _lib.Context(0)
    _check_ig(ctx, N=4096, D=11, M=8192, Nb=50, Np=400, seed=4, kind="fabolas")
    ctx.close()


def test_entropy_search_front_end_emulated(emu_ctx):
    """robo_amd.fmin.entropy_search, the assertions of the reference's test_fmin_interface.py (x_opt in bounds)"""
    from robo_amd.fmin import entropy_search
    r = entropy_search(lambda x: float((x[0] - 0.3) ** 2), np.zeros(1), np.ones(1), num_iterations=4, model="gp",
                       rng=np.random.RandomState(0), n_candidates=40, n_representer=6, n_outcomes=12)
    assert 0.0 <= r["x_opt"][0] <= 1.0 and len(r["X"]) == 4


@pytest.mark.gpu
def test_entropy_search_and_fabolas_front_ends_gpu():
    """the reference's test/test_fmin: x_opt inside the bounds for entropy_search (gp, gp_mcmc) and fabolas
    (test_fmin_interface.py:18-91, test_fabolas.py:24-36), at the reference's Nb=50 / Np=400"""
    _lib.use_library(None)
    from robo_amd.fmin import entropy_search, fabolas
    f = lambda x: float((x[0] - 0.3) ** 2)
    for model in ("gp", "gp_mcmc"):
        r = entropy_search(f, np.zeros(1), np.ones(1), num_iterations=6, model=model, rng=np.random.RandomState(0),
                           n_candidates=2000, chain_length=10, burnin_steps=10)
        assert 0.0 <= r["x_opt"][0] <= 1.0 and len(r["X"]) == 6

    def obj(x, s):
        return float((x[0] - 0.5) ** 2 + 1.0 / s + 0.05), float(s) / 100.0

    r = fabolas(obj, np.zeros(1), np.ones(1), s_min=10, s_max=1000, n_init=3, num_iterations=10, subsets=[64, 16],
                burnin=5, chain_length=5, n_hypers=10, rng=np.random.RandomState(1), n_candidates=1000)
    assert 0.0 <= r["x_opt"][0] <= 1.0 and len(r["X"]) == 10 and len(r["c"]) == 10

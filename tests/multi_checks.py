"""Checks of the single-process multi-device entry points, shared by the CPU suite (emulated devices,
tests/test_multi_device.py) and the GPU suite (two contexts on the MI355X).  Every expected value comes from the
SINGLE-device entry points on the same inputs: sharding must not change a bit of what a shard computes, and the
reductions are specified exactly (np.argmax tie-break; partial sums added in device order)."""
import numpy as np

from robo_amd import _lib


def _problem(N=90, D=3, M=301, seed=5, kind="matern52"):
    rs = np.random.RandomState(seed)
    X = rs.rand(N, D)
    y = np.sin(3 * X.sum(axis=1)) + 0.05 * rs.randn(N)
    theta = np.concatenate([[0.1], np.log(0.3 * D) + 0.2 * rs.randn(D), [np.log(1e-2)]])
    Xc = rs.rand(M, D)
    return X, y, theta, Xc


def _replicas(multi, X, y, theta, kind="matern52"):
    gps = [_lib.DeviceGP(c, kind, X.shape[0], X.shape[1]) for c in multi.ctxs]
    multi.set_data(gps, X, y)
    ll = multi.fit(gps, theta, float(y.mean()))
    return gps, ll


def check_candidate_shard(devices, sizes=((90, 3, 301), (40, 2, 2))):
    multi = _lib.multi_for(devices)
    for N, D, M in sizes:                        # M = 2 with 3 devices: an empty shard
        X, y, theta, Xc = _problem(N, D, M)
        Xc[M // 2] = Xc[0]                       # a duplicate row: equal values in two shards -> the lower index wins
        gps, ll = _replicas(multi, X, y, theta)
        single = _lib.DeviceGP(multi.ctxs[0], "matern52", N, D)
        single.set_data(X, y)
        assert single.fit(theta, float(y.mean())) == ll
        for kind, par in (("ei", 0.0), ("log_ei", 0.01), ("lcb", 1.0)):
            eta = float(y.min())
            vals, mx, am, fl = single.acq(kind, par, eta, Xc)
            shards = _lib.CandidateShards.split(multi.ctxs, Xc)
            try:
                v2, mx2, am2, owner, fl2 = multi.acq(gps, kind, par, eta, shards, want_values=True)
                np.testing.assert_array_equal(v2, vals)
                assert (mx2, am2, fl2) == (mx, am, fl) and am2 == int(np.argmax(vals))
                b, e = _lib.shard_range(M, owner, multi.n)
                assert b <= am2 < e
                np.testing.assert_array_equal(shards.owner_point(owner, am2), Xc[am2])
                _, mx3, am3, _, _ = multi.acq(gps, kind, par, eta, shards, want_values=False)
                assert (mx3, am3) == (mx, am)
            finally:
                shards.close()
        # constant acquisition values (LCB of identical points): the FIRST index, as np.argmax
        Xsame = np.tile(Xc[:1], (M, 1))
        shards = _lib.CandidateShards.split(multi.ctxs, Xsame)
        try:
            assert multi.acq(gps, "lcb", 1.0, 0.0, shards)[2] == 0
        finally:
            shards.close()
        for g in gps + [single]:
            g.close()


def check_sample_shard(devices, N=70, D=3, M=140, S=5):
    multi = _lib.multi_for(devices)
    G = multi.n
    X, y, theta, Xc = _problem(N, D, M, seed=9)
    rs = np.random.RandomState(3)
    thetas = theta[None, :] + 0.3 * rs.randn(S, theta.size)
    mean_c = float(y.mean())
    etas = float(y.min()) + 0.01 * np.arange(S)
    ctx0 = multi.ctxs[0]
    single = [_lib.DeviceGP(ctx0, "matern52", N, D) for _ in range(S)]
    single[0].set_data(X, y)
    ll_ref, st = _lib.fit_batch(single, thetas, mean_c)
    assert np.all(st == _lib.OK)
    cand0 = _lib.Candidates(ctx0, Xc)
    groups, eta_groups = [], []
    for g in range(G):
        b, e = _lib.shard_range(S, g, G)
        grp = [_lib.DeviceGP(multi.ctxs[g], "matern52", N, D) for _ in range(b, e)]
        if grp:
            grp[0].set_data(X, y)
        groups.append(grp)
        eta_groups.append(etas[b:e])
    ll, st = multi.fit_batch(groups, thetas, mean_c)
    assert np.all(st == _lib.OK)
    np.testing.assert_array_equal(ll, ll_ref)
    cands = [_lib.Candidates(multi.ctxs[g], Xc) if (groups[g] or g == 0) else None for g in range(G)]
    try:
        for kind, par in (("log_ei", 0.0), ("ei", 0.0)):
            # expected: per-device partial sums (single-device entry point), added in device order, divided by S
            total = None
            for g in range(G):
                b, e = _lib.shard_range(S, g, G)
                if e == b:
                    continue
                part, _, _, _ = _lib.acq_marginal(single[b:e], kind, par, etas[b:e], cand0, reduce="sum")
                total = part.copy() if total is None else total + part
            expect = total / S
            vals, mx, am, fl = multi.acq_marginal(groups, kind, par, eta_groups, cands)
            np.testing.assert_array_equal(vals, expect)
            assert am == int(np.argmax(expect)) and mx == expect[am]
            ref, _, am1, _ = _lib.acq_marginal(single, kind, par, etas, cand0)       # sample-order accumulation
            np.testing.assert_allclose(vals, ref, rtol=1e-13, atol=1e-300)
            _, mx2, am2, _ = multi.acq_marginal(groups, kind, par, eta_groups, cands, want_values=False)
            assert (mx2, am2) == (mx, am)
        # the mixture posterior of GaussianProcessMCMC.predict: identical to the single-device mixture
        m1, v1 = _lib.predict_mixture(single, cand0)
        m2, v2 = multi.predict_mixture(groups, cands)
        np.testing.assert_array_equal(m2, m1)
        np.testing.assert_array_equal(v2, v1)
    finally:
        for c in cands + [cand0]:
            if c is not None:
                c.close()
        for g in single + [g for grp in groups for g in grp]:
            g.close()


def check_fits_and_mixture(devices, N=80, D=3, S=7):
    multi = _lib.multi_for(devices)
    X, y, theta, _ = _problem(N, D, 10, seed=11)
    gps, ll = _replicas(multi, X, y, theta)
    single = _lib.DeviceGP(multi.ctxs[0], "matern52", N, D)
    single.set_data(X, y)
    assert single.fit(theta, float(y.mean())) == ll
    for g in gps:
        np.testing.assert_array_equal(g.factor(), single.factor())
    thetas = theta[None, :] + 0.3 * np.random.RandomState(1).randn(S, theta.size)
    thetas[2, 1] = np.nan                               # a walker outside the library's domain: reported per theta
    ll1, st1 = single.loglik_batch(thetas, float(y.mean()))
    ll2, st2 = multi.loglik_batch(gps, thetas, float(y.mean()))
    np.testing.assert_array_equal(st2, st1)
    np.testing.assert_array_equal(ll2, ll1)
    # a not-positive-definite fit is the same error on every replica
    bad = theta.copy()
    bad[0] = 710.0
    try:
        multi.fit(gps, bad, float(y.mean()))
        raise AssertionError("expected LinAlgError")
    except np.linalg.LinAlgError:
        pass
    for g in gps + [single]:
        g.close()


def check_per_cost_shard(devices, N=60, D=4, M=97, Nb=10, Np=30):
    """information gain per unit cost (robo_ig_eval_per_cost_cand_multi) == the single-device call on the whole batch"""
    from oracle import gp_oracle as O
    from oracle import ig_oracle as IG
    from robo_amd.util import epmgp
    multi = _lib.multi_for(devices)
    rs = np.random.RandomState(21)
    X = rs.rand(N, D)
    X[:, -1] = (1.0 - X[:, -1]) ** 2
    y = np.sin(3 * X.sum(axis=1)) + 0.1 * rs.randn(N)
    c = 0.5 * X[:, -1] + 0.1 * rs.randn(N)
    theta = np.concatenate([[0.0], np.log(0.3 * (D - 1)) + 0.2 * rs.randn(D - 1), [-0.5, 0.3], [np.log(1e-2)]])
    theta_c = theta + 0.1
    Xc = rs.rand(M, D)
    zb = rs.rand(Nb, D)
    zb[:, -1] = 0.0
    lmb = rs.randn(Nb)
    ogp = O.OracleGP("fabolas", theta, normalize_input=False)
    ogp.train(X, y)
    mu_b, var_b = ogp.predict(zb, full_cov=True)
    logP, dMu, dSig, dMM = epmgp.joint_min(mu_b, var_b, with_derivatives=True)
    ep = _lib.EPState(logP, lmb, IG.outcome_quantiles(Np), dMu, dSig, dMM)
    sn2 = float(np.exp(theta[-1]))
    gps, _ = _replicas(multi, X, y, theta, kind="fabolas")
    cgps = [_lib.DeviceGP(cx, "fabolas", N, D) for cx in multi.ctxs]
    multi.set_data(cgps, X, c)
    multi.fit(cgps, theta_c, float(c.mean()))
    ctx0 = multi.ctxs[0]
    cand, ccand, rep = _lib.Candidates(ctx0, Xc), _lib.Candidates(ctx0, Xc), _lib.Candidates(ctx0, zb)
    vals, mx, am = _lib.ig_eval_per_cost(gps[0], cand, rep, ep, sn2, cgps[0], ccand, 0.25)
    shards, cshards = _lib.CandidateShards.split(multi.ctxs, Xc), _lib.CandidateShards.split(multi.ctxs, Xc)
    reps = [_lib.Candidates(cx, zb) for cx in multi.ctxs]
    try:
        v2, mx2, am2, owner = multi.ig_per_cost(gps, shards, reps, ep, sn2, cgps, cshards, 0.25, want_values=True)
        np.testing.assert_array_equal(v2, vals)
        assert (mx2, am2) == (mx, am) and am == int(np.argmax(vals))
        assert _lib.shard_range(M, owner, multi.n)[0] <= am2 < _lib.shard_range(M, owner, multi.n)[1]
        # the plain information gain (robo_ig_eval_cand_multi) on the same shards
        g_vals, g_mx, g_am = _lib.ig_eval(gps[0], cand, rep, ep, sn2)
        v3, mx3, am3, _ = multi.ig(gps, shards, reps, ep, sn2, want_values=True)
        np.testing.assert_array_equal(v3, g_vals)
        assert (mx3, am3) == (g_mx, g_am)
        assert multi.ig(gps, shards, reps, ep, sn2, want_values=False)[1:3] == (g_mx, g_am)
    finally:
        for h in [cand, ccand, rep] + reps:
            h.close()
        shards.close()
        cshards.close()
        for g in gps + cgps:
            g.close()


def check_failing_device(devices):
    """a device whose local half fails (an unfitted replica): the call returns that failure after ALL devices finished
    -- there is no collective to hang in -- and the next call on the same objects works"""
    import pytest
    multi = _lib.multi_for(devices)
    X, y, theta, Xc = _problem(60, 3, 64, seed=2)
    gps, _ = _replicas(multi, X, y, theta)
    broken = _lib.DeviceGP(multi.ctxs[1], "matern52", 60, 3)       # never fitted
    shards = _lib.CandidateShards.split(multi.ctxs, Xc)
    try:
        with pytest.raises(Exception, match="trained first"):
            multi.acq([gps[0], broken] + gps[2:], "ei", 0.0, float(y.min()), shards)
        vals, mx, am, _ = gps[0].acq("ei", 0.0, float(y.min()), Xc)
        assert multi.acq(gps, "ei", 0.0, float(y.min()), shards)[1:3] == (mx, am)
        # a handle on the wrong context is refused before anything is launched
        with pytest.raises(ValueError, match="another context"):
            multi.acq([gps[1], gps[0]] + gps[2:], "ei", 0.0, float(y.min()), shards)
    finally:
        shards.close()
        broken.close()
        for g in gps:
            g.close()


# ---- the front ends on several devices of one process ----------------------------------------------------------------------
def branin(x):
    a, b, c, r, s, t = 1.0, 5.1 / (4 * np.pi ** 2), 5.0 / np.pi, 6.0, 10.0, 1.0 / (8 * np.pi)
    return a * (x[1] - b * x[0] ** 2 + c * x[0] - r) ** 2 + s * (1 - t) * np.cos(x[0]) + s


class Counted(object):
    def __init__(self, f):
        self.f, self.calls = f, 0

    def __call__(self, *a):
        self.calls += 1
        return self.f(*a)


def run_bo(devices=None, n_gpus=None, num_iterations=8, **kw):
    """robo_amd.fmin.bayesian_optimization on Branin with fixed seeds -> (X trajectory, objective calls)"""
    from robo_amd.fmin import bayesian_optimization
    f = Counted(branin)
    np.random.seed(11)
    res = bayesian_optimization(f, np.array([-5.0, 0.0]), np.array([10.0, 15.0]), num_iterations=num_iterations, n_init=3,
                                rng=np.random.RandomState(11), n_gpus=n_gpus, devices=devices, **kw)
    return np.array(res["X"]), f.calls


def check_front_end_trajectories(device_lists, num_iterations=8, mcmc=dict(chain_length=4, burnin_steps=6)):
    """bayesian_optimization(devices=...) == the one-device run, point for point, for model_type gp (candidate shard over
    replicas) and gp_mcmc (sample shard); the objective is evaluated once per iteration"""
    for kw in (dict(model_type="gp", acquisition_func="ei", maximizer="random"),
               dict(model_type="gp_mcmc", acquisition_func="log_ei", maximizer="random", **mcmc)):
        X1, calls1 = run_bo(None, num_iterations=num_iterations, **kw)
        for devices in device_lists:
            XG, calls = run_bo(devices, num_iterations=num_iterations, **kw)
            np.testing.assert_array_equal(XG, X1, err_msg="%r on devices %r" % (kw["model_type"], devices))
            assert calls == calls1 == num_iterations


def check_walker_shard(devices, N=60, D=3, n_hypers=10):
    """GaussianProcessMCMC(devices=...) with the walkers of every ensemble half-step split over the devices (host sampler
    around robo_gp_loglik_batch_multi) against the single-device chain (robo_gp_mcmc_run on the device): the same hyper-
    parameter samples (per-theta likelihoods are bit-identical; the host and the device sampler agree to rounding)"""
    from robo_amd.kernels import Matern52Kernel
    from robo_amd.models import GaussianProcessMCMC
    from robo_amd.priors import DefaultPrior
    rs = np.random.RandomState(4)
    X = rs.rand(N, D)
    y = np.sin(3 * X.sum(axis=1))

    def build(devs, walker_min):
        kernel = 2 * Matern52Kernel(np.ones(D), ndim=D)
        m = GaussianProcessMCMC(kernel, prior=DefaultPrior(len(kernel) + 1, rng=np.random.RandomState(5)), n_hypers=n_hypers,
                                chain_length=4, burnin_steps=6, rng=np.random.RandomState(7), lower=np.zeros(D),
                                upper=np.ones(D), devices=devs)
        m.walker_shard_min_n = walker_min
        m.train(X, y)
        return m
    one, many = build(None, 10 ** 9), build(devices, 1)
    assert len(many.walker_gps) == len(devices) - 1
    np.testing.assert_allclose(np.array(many.hypers), np.array(one.hypers), rtol=1e-9)
    Xc = rs.rand(50, D)
    for a, b in zip(many.predict(Xc), one.predict(Xc)):
        np.testing.assert_allclose(a, b, rtol=1e-8, atol=1e-12)
    # a later train(do_optimize=False) on MORE data (the solver's train_interval > 1, Fabolas keeping its samples): the
    # walker handles of devices 1.. must see the new data too -- a direct likelihood call afterwards scores every theta
    # against it (round-5 advice: they kept the previous data set and the batch silently mixed two)
    thetas = np.array(one.hypers) + 0.01 * rs.randn(n_hypers, D + 2)       # (inside the prior's support)
    X2 = np.vstack([X, rs.rand(7, D)])
    y2 = np.sin(3 * X2.sum(axis=1))
    many.train(X2, y2, do_optimize=False)
    one.train(X2, y2, do_optimize=False)
    ll = one.loglikelihood_batch(thetas)
    assert np.all(np.isfinite(ll))
    np.testing.assert_array_equal(many.loglikelihood_batch(thetas), ll)

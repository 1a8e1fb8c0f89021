"""Parity checks of the HIP path against the oracle and the golden fixtures.

The same functions run in two settings:
  * tests/test_gpu_parity.py  (-m gpu): the real librobo_hip.so on an MI355X, through the C ABI;
  * tests/test_emu_logic.py   (CPU)   : the g++-interpreted build of the same sources (tests/hipemu)
    -- checks index arithmetic / host logic only, proves nothing about the hardware.
Tolerances are the stated fp64 tolerances of tests/_tol.py.
"""
import os

import numpy as np

from _tol import ACQ_RTOL, LOGLIK_RTOL, MU_ATOL, MU_RTOL, VAR_ATOL_REL_AMP, assert_logei_close
from make_golden import golden_inputs, mcmc_inputs
from oracle import gp_oracle as O
from robo_amd import _lib

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def oracle_gp(inp):
    gp = O.OracleGP(inp["kind"], inp["theta"], normalize_output=inp["nout"], lower=inp["lower"],
                    upper=inp["upper"])
    gp.train(inp["X"], inp["y"])
    return gp


def device_gp(ctx, ogp, inp):
    """device GP on the oracle's normalised data (the C-ABI sees what george would see)"""
    g = _lib.DeviceGP(ctx, inp["kind"], ogp.X.shape[0], ogp.X.shape[1])
    g.set_data(ogp.X, ogp.y)
    if inp["nout"]:
        g.set_output_transform(ogp.y_mean, ogp.y_std)
    return g


def check_case(ctx, name, full=True):
    inp = golden_inputs(name)
    gold = load(name)
    ogp = oracle_gp(inp)
    g = device_gp(ctx, ogp, inp)
    amp = np.exp(inp["theta"][0]) * (ogp.y_std ** 2 if inp["nout"] else 1.0)
    N = ogp.X.shape[0]

    if full:
        # K1 gram
        K = g.gram(inp["theta"])
        Ko = O.kernel_matrix(inp["kind"], inp["theta"][:-1], ogp.X) + (ogp.noise + O.JITTER) * np.eye(N)
        np.testing.assert_allclose(K, Ko, rtol=1e-13, atol=1e-15)
    # K2/K3 fit + log-likelihood
    ll = g.fit(inp["theta"], ogp.mean)
    np.testing.assert_allclose(ll, float(gold["loglik"]), rtol=LOGLIK_RTOL)
    if full:
        L = g.factor()
        np.testing.assert_allclose(L, ogp.L, rtol=0, atol=1e-11 * np.abs(ogp.L).max())
    # K4/K5 posterior
    Xcn = O.zero_one_normalization(inp["Xc"], inp["lower"], inp["upper"])[0]
    mu, var = g.predict(Xcn)
    np.testing.assert_allclose(mu, gold["mu"], rtol=MU_RTOL, atol=MU_ATOL * max(1.0, np.abs(gold["mu"]).max()))
    np.testing.assert_allclose(var, gold["var"], rtol=0, atol=VAR_ATOL_REL_AMP * amp)
    assert np.all(var >= O.EPS)
    # K6 acquisition + argmax.  Two comparisons per function:
    #  (a) element-wise kernel: device values vs the oracle formula on the DEVICE's own (mu, var)
    #      -- tight, isolates the Phi/phi/log arithmetic from the posterior's conditioning;
    #  (b) end to end vs the REFERENCE classes' outputs in the fixture, where the acquisition is
    #      well conditioned w.r.t. (mu, var) (|z| < 8): abs tolerance = the posterior tolerance.
    eta = float(gold["eta"])
    z = (eta - gold["mu"]) / np.sqrt(gold["var"])
    well = np.abs(z) < 8
    scale = max(1.0, np.abs(gold["mu"]).max())
    own = {"ei": lambda par: O.ei(mu, var, eta, par), "pi": lambda par: O.pi(mu, var, eta, par),
           "lcb": lambda par: O.lcb(mu, var, par)}
    for kind, par, key in (("ei", 0.0, "ei"), ("pi", 0.0, "pi"), ("lcb", 1.0, "lcb"), ("ei", 0.3, "ei_par"),
                           ("lcb", 2.5, "lcb_par")):
        vals, mx, am, flags = g.acq(kind, par, eta, Xcn)
        np.testing.assert_allclose(vals, own[kind](par), rtol=1e-10, atol=1e-14 * scale)       # (a)
        ref = gold[key]
        np.testing.assert_allclose(vals[well], ref[well], rtol=ACQ_RTOL, atol=1e-9 * scale)   # (b)
        assert am == int(np.argmax(vals)) and mx == vals[am]
        if key in ("ei", "pi", "lcb"):
            gap = float(gold["gap_" + key])
            want = int(gold["argmax_" + key])
            # argmax index must be bit-exact unless the oracle's own top-2 gap is below tolerance
            assert am == want or gap <= ACQ_RTOL * abs(ref[want]), (kind, am, want, gap)
        assert not (flags & _lib.FLAG_NAN)
    vals, mx, am, _ = g.acq("log_ei", 0.0, eta, Xcn)
    # LogEI amplifies d(mu,var): compare on the GPU's own moments with the oracle formula (tight)
    # and against the reference fixture (loose where |z| is large)
    assert_logei_close(vals, O.log_ei_vec(mu, var, eta), (eta - mu) / np.sqrt(var), rtol=1e-11, tail_rtol=1e-7)
    core = np.abs(z) < 8
    np.testing.assert_allclose(vals[core], gold["log_ei"][core], rtol=1e-6, atol=1e-9)
    want = int(gold["argmax_log_ei"])
    assert am == want or float(gold["gap_log_ei"]) <= 1e-6 * abs(gold["log_ei"][want])
    if full and "cov33" in gold.files:
        _, cov = g.predict_cov(Xcn[:33])
        eps = np.finfo(np.float64).eps
        np.testing.assert_allclose(np.clip(cov, eps, np.inf), gold["cov33"], rtol=0, atol=VAR_ATOL_REL_AMP * amp)
    g.close()


def check_device_chain(ctx, cases=(("matern52", 150, 3, 10, 14), ("rbf", 40, 2, 8, 10), ("matern52", 100, 4, 12, 9),
                                   ("fabolas", 50, 3, 12, 8), ("fabolas", 60, 3, 12, 8, "env"),
                                   ("fabolas", 170, 4, 14, 6, "env"))):
    """robo_gp_mcmc_run (the whole stretch-move chain on the device) against the host sampler around the batched
    likelihood with the same RandomState: same accept decisions, positions and log-probabilities to rounding, the random
    stream ends in the same state; against the CPU oracle's log-probability through the same sampler; walkers outside
    the reference's |theta| <= 20 bounds and outside the prior's support (-inf); the model class end to end."""
    from robo_amd.util.ensemble_sampler import EnsembleSampler
    from robo_amd.priors import DefaultPrior, EnvPrior
    for case in cases:
        kind, N, D, k, steps = case[:5]
        prior_name = case[5] if len(case) > 5 else "default"
        rs = np.random.RandomState(61)
        X = rs.rand(N, D)
        y = np.sin(3 * X.sum(axis=1)) + 0.05 * rs.randn(N)
        P = O.n_kernel_params(kind, D) + 1          # fabolas: the two parameters of the linear fidelity kernel as well
        mean = float(np.mean(y))
        g = _lib.DeviceGP(ctx, kind, N, D)
        g.set_data(X, y)
        if prior_name == "env":
            # FabolasGPMCMC's prior (robo/fmin/fabolas.py:120-127): tophat on the D - 1 length scales only, a normal PDF
            # term per regression parameter, lognormal(-2, 1) amplitude, horseshoe(0.001) noise
            prior = EnvPrior(P, n_ls=D - 1, n_lr=2, rng=np.random.RandomState(62))
            par = (2, [prior.ln_prior.mean, prior.ln_prior.sigma, prior.tophat.min, prior.tophat.max,
                       prior.horseshoe.scale, prior.n_ls, prior.n_lr, prior.bayes_lin_prior.mean,
                       prior.bayes_lin_prior.sigma])
            prior.lnprob_batch = lambda T: np.array([prior.lnprob(t) for t in np.atleast_2d(T)])
        else:
            prior = DefaultPrior(P, rng=np.random.RandomState(62))
            par = (1, [prior.ln_prior.mean, prior.ln_prior.sigma, prior.tophat.min, prior.tophat.max,
                       prior.horseshoe.scale])
        p0 = prior.sample_from_prior(k)
        if prior_name == "env":
            p0[:, -1] = np.clip(p0[:, -1], -14.0, None)      # horseshoe(0.001) samples reach exp(-30): keep K factorable
        p0[1, 2] = 23.0          # outside |theta| <= 20: -inf at the start, moves in through a partner
        p0[2, 1] = 5.0           # outside the tophat: -inf prior

        def lnprob_host(thetas):
            thetas = np.atleast_2d(thetas)
            out = np.full(thetas.shape[0], -np.inf)
            ok = ~np.any((-20 > thetas) + (thetas > 20), axis=1)
            if np.any(ok):
                ll, st = g.loglik_batch(thetas[ok], mean)
                out[ok] = np.where(st == _lib.OK, ll, -np.inf) + prior.lnprob_batch(thetas[ok])
            return out

        def lnprob_oracle(thetas):
            out = []
            for th in np.atleast_2d(thetas):
                if np.any((-20 > th) + (th > 20)):
                    out.append(-np.inf)
                    continue
                try:
                    L = O.gp_compute(kind, th, X, np.float64)
                    out.append(O.gp_log_likelihood(L, y, mean) + prior.lnprob(th))
                except np.linalg.LinAlgError:
                    out.append(-np.inf)
            return np.array(out)

        def run(**kw):
            smp = EnsembleSampler(k, P, **kw)
            smp.random_state = np.random.RandomState(63).get_state()
            pos, lnp, _ = smp.run_mcmc(p0, steps)
            pos2, lnp2, state = smp.run_mcmc(pos, 3, lnprob0=lnp)       # continue with a given lnprob0
            return smp.chain, smp.lnprobability, smp.naccepted.copy(), pos2, lnp2, state

        dev = run(lnprob_batch=lnprob_host, device_chain=lambda p, lnp, n, uz, pa, ua, a: g.mcmc_run(mean, par, p, lnp, n, uz, pa, ua, a))
        if N <= 254 and kind != "fabolas":
            # one- and two-block problems: the half-step fused into ONE launch (default; one tile group below 64 points,
            # three above; N >= 128: mcmc_block2_step_kernel) against the launch-per-phase form -- same accept
            # decisions; likelihoods bit-identical on the emulator, within an ulp on the MI355X (fused-multiply-add
            # contraction is decided per kernel)
            try:
                if N > 126:      # the two-block form is an option (mcmc_block_step = 3), not the default: run it explicitly
                    ctx.set_tuning("mcmc_block_step", 3)
                    dev = run(lnprob_batch=lnprob_host,
                              device_chain=lambda p, lnp, n, uz, pa, ua, a: g.mcmc_run(mean, par, p, lnp, n, uz, pa, ua, a))
                ctx.set_tuning("mcmc_block_step", 0)
                dev4 = run(lnprob_batch=lnprob_host,
                           device_chain=lambda p, lnp, n, uz, pa, ua, a: g.mcmc_run(mean, par, p, lnp, n, uz, pa, ua, a))
            finally:
                ctx.set_tuning("mcmc_block_step", None)
            np.testing.assert_array_equal(dev[2], dev4[2])
            for a_, b_ in ((dev[0], dev4[0]), (dev[3], dev4[3])):
                np.testing.assert_allclose(a_, b_, rtol=1e-12, atol=1e-12)
            fin4 = np.isfinite(dev4[1])
            assert np.array_equal(fin4, np.isfinite(dev[1]))
            np.testing.assert_allclose(dev[1][fin4], dev4[1][fin4], rtol=1e-13, atol=0)
            if "hipemu" in ctx.name:
                np.testing.assert_array_equal(dev[1][fin4], dev4[1][fin4])
        if N > 126:
            # multi-block factors: likelihood terms + accept test in ONE launch (mcmc_fused_tail, default) against the three
            # launches of the factorisation's tail + the accept kernel: the same operations in the same order, the same chain
            ctx.set_tuning("mcmc_fused_tail", 0)
            try:
                dev3 = run(lnprob_batch=lnprob_host,
                           device_chain=lambda p, lnp, n, uz, pa, ua, a: g.mcmc_run(mean, par, p, lnp, n, uz, pa, ua, a))
            finally:
                ctx.set_tuning("mcmc_fused_tail", None)
            np.testing.assert_array_equal(dev[2], dev3[2])                       # same accept decisions per walker
            np.testing.assert_array_equal(dev[0], dev3[0])                       # hence the same walkers, bit for bit
            np.testing.assert_array_equal(dev[3], dev3[3])
            fin3 = np.isfinite(dev3[1])
            assert np.array_equal(fin3, np.isfinite(dev[1]))
            # the log-probabilities: bit-identical through the interpreter, within an ulp on the MI355X (whether zi * zi and
            # the sums around it contract is decided per kernel)
            np.testing.assert_allclose(dev[1][fin3], dev3[1][fin3], rtol=1e-13, atol=0)
            if "hipemu" in ctx.name:
                np.testing.assert_array_equal(dev[1][fin3], dev3[1][fin3])
        host = run(lnprob_batch=lnprob_host)
        orc = run(lnprob_batch=lnprob_oracle)
        assert dev[0].shape == (k, steps + 3, P) and np.all(np.isfinite(dev[0]))
        for ref, tol in ((host, 1e-9), (orc, 1e-7)):
            np.testing.assert_array_equal(dev[2], ref[2])                       # same accept decisions per walker
            np.testing.assert_allclose(dev[0], ref[0], rtol=tol, atol=tol)
            fin = np.isfinite(ref[1])
            assert np.array_equal(fin, np.isfinite(dev[1]))
            np.testing.assert_allclose(dev[1][fin], ref[1][fin], rtol=0, atol=1e-7 * max(1.0, np.abs(ref[1][fin]).max()))
            np.testing.assert_allclose(dev[3], ref[3], rtol=tol, atol=tol)
        assert dev[2].sum() > 0 and np.isneginf(dev[1][1, 0]) or dev[2][1] > 0   # the out-of-bounds walker started at -inf
        for a_, b_ in zip(dev[5][1:], host[5][1:]):
            assert np.array_equal(np.asarray(a_), np.asarray(b_))                # the random stream ends in the same state
        # no prior
        d0 = EnsembleSampler(k, P, lnprob_batch=lnprob_host,
                             device_chain=lambda p, lnp, n, uz, pa, ua, a: g.mcmc_run(mean, None, p, lnp, n, uz, pa, ua, a))
        d0.random_state = np.random.RandomState(64).get_state()
        q0 = np.clip(p0, -3, 1.5)
        d0.run_mcmc(q0, 5)

        def lnprob_noprior(thetas):
            ll, st = g.loglik_batch(np.atleast_2d(thetas), mean)
            return np.where(st == _lib.OK, ll, -np.inf)
        h0 = EnsembleSampler(k, P, lnprob_batch=lnprob_noprior)
        h0.random_state = np.random.RandomState(64).get_state()
        h0.run_mcmc(q0, 5)
        np.testing.assert_allclose(d0.chain, h0.chain, rtol=1e-9, atol=1e-9)
        np.testing.assert_array_equal(d0.naccepted, h0.naccepted)
        # zero steps: only the start positions are evaluated; emcee's error for a +inf start (the horseshoe term at
        # log-noise == 0 is +inf, priors.py) comes back as the ValueError the host sampler raises
        e0 = np.empty((0, 2, k // 2))
        pos_z, lnp_z, ch_z, _, acc_z = g.mcmc_run(mean, par, q0, None, 0, e0, e0.astype(np.int32), e0)
        np.testing.assert_array_equal(pos_z, q0)
        assert ch_z.shape == (k, 0, P) and not acc_z.any()
        fin = np.isfinite(lnp_z)
        np.testing.assert_allclose(lnp_z[fin], lnprob_host(q0)[fin], rtol=0, atol=1e-7 * np.abs(lnp_z[fin]).max())
        q_inf = q0.copy()
        q_inf[0, -1] = 0.0
        try:
            g.mcmc_run(mean, par, q_inf, None, 0, e0, e0.astype(np.int32), e0)
            raise AssertionError("a +inf start must be refused")
        except ValueError as e:
            assert "+inf" in str(e), str(e)
        g.close()
    # the model class: device chain (default) == host sampler (ROBO_MCMC_HOST=1)
    from robo_amd.kernels import Matern52Kernel
    from robo_amd.models.gaussian_process_mcmc import GaussianProcessMCMC
    rs = np.random.RandomState(65)
    X = rs.rand(60, 2) * 3 - 1
    y = np.sin(X[:, 0]) * np.cos(2 * X[:, 1])
    hyp = {}
    for mode in ("0", "1", "declined"):
        # "declined": half an ensemble does not fit the batch workspace -> robo_gp_mcmc_run returns ROBO_BAD_SHAPE and the
        # sampler continues on the host with the random numbers it had already drawn
        os.environ["ROBO_MCMC_HOST"] = "1" if mode == "1" else "0"
        if mode == "declined":
            _lib.default_context().set_tuning("ws_bytes", 200000)
        try:
            kernel = 2 * Matern52Kernel(np.ones([2]), ndim=2)
            m = GaussianProcessMCMC(kernel, prior=DefaultPrior(len(kernel) + 1, rng=np.random.RandomState(66)),
                                    n_hypers=8, chain_length=12, burnin_steps=10, rng=np.random.RandomState(67),
                                    lower=-np.ones(2), upper=2 * np.ones(2), device=None)
            m.train(X, y)
            m.train(X, y)                     # burned: chain only, from p0
            hyp[mode] = (np.array(m.hypers), m.predict(X[:7] + 0.1))
        finally:
            os.environ.pop("ROBO_MCMC_HOST", None)
            _lib.default_context().set_tuning("ws_bytes", None)
    np.testing.assert_array_equal(hyp["declined"][0], hyp["1"][0])
    np.testing.assert_allclose(hyp["0"][0], hyp["1"][0], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(hyp["0"][1][0], hyp["1"][1][0], rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(hyp["0"][1][1], hyp["1"][1][1], rtol=1e-6, atol=1e-10)
    # FabolasGPMCMC with EnvPrior (config 4's model, robo/fmin/fabolas.py:104-135): device chain == host sampler
    from robo_amd.kernels import FabolasKernel
    from robo_amd.models.fabolas_gp import FabolasGPMCMC
    rs = np.random.RandomState(68)
    Xf = np.concatenate([rs.rand(70, 2) * 3 - 1, rs.rand(70, 1) * 0.9 + 0.1], axis=1)
    yf = np.sin(Xf[:, 0]) * np.cos(2 * Xf[:, 1]) + 0.3 * (1 - Xf[:, 2]) ** 2
    hyp = {}
    for mode in ("0", "1"):
        os.environ["ROBO_MCMC_HOST"] = mode
        try:
            kernel = FabolasKernel(3, metric=0.01)
            m = FabolasGPMCMC(kernel, basis_func=lambda s_: (1 - s_) ** 2,
                              prior=EnvPrior(len(kernel) + 1, n_ls=2, n_lr=2, rng=np.random.RandomState(69)),
                              n_hypers=12, chain_length=10, burnin_steps=8, rng=np.random.RandomState(70),
                              lower=-np.ones(2), upper=2 * np.ones(2), device=None)
            chain = m._device_chain()
            assert (chain is None) == (mode == "1")          # EnvPrior is a prior the library evaluates itself
            m.train(Xf, yf)
            m.train(Xf, yf)
            Xt = np.concatenate([Xf[:7, :2] + 0.1, np.ones((7, 1))], axis=1)
            hyp[mode] = (np.array(m.hypers), m.predict(Xt))
        finally:
            os.environ.pop("ROBO_MCMC_HOST", None)
    np.testing.assert_allclose(hyp["0"][0], hyp["1"][0], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(hyp["0"][1][0], hyp["1"][1][0], rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(hyp["0"][1][1], hyp["1"][1][1], rtol=1e-6, atol=1e-10)


def check_mcmc_marginal(ctx):
    inp = mcmc_inputs()
    gold = load("mcmc_marginal")
    Xn = O.zero_one_normalization(inp["X"], inp["lower"], inp["upper"])[0]
    Xcn = O.zero_one_normalization(inp["Xc"], inp["lower"], inp["upper"])[0]
    y = inp["y"]
    gps = []
    for th in inp["thetas"]:
        g = _lib.DeviceGP(ctx, inp["kind"], Xn.shape[0], Xn.shape[1])
        g.set_data(Xn, y)
        g.fit(th, float(np.mean(y)))
        gps.append(g)
    ll, st = gps[0].loglik_batch(inp["thetas"], float(np.mean(y)))
    assert np.all(st == _lib.OK)
    np.testing.assert_allclose(ll, gold["loglik_s"], rtol=LOGLIK_RTOL)
    gps[0].fit(inp["thetas"][0], float(np.mean(y)))     # loglik_batch leaves the GP unfitted
    cand = _lib.Candidates(ctx, Xcn)
    eta = float(y.min())
    for s, g in enumerate(gps):
        mu, var = g.predict(cand)
        np.testing.assert_allclose(mu, gold["mu_s"][s], rtol=MU_RTOL, atol=MU_ATOL)
        np.testing.assert_allclose(var, gold["var_s"][s], rtol=0, atol=VAR_ATOL_REL_AMP * np.exp(inp["thetas"][s][0]))
    # K7 mixture: device == NumPy on the device's own per-sample moments (bit for bit), and == fixture
    per_mu = np.array([g.predict(cand)[0] for g in gps])
    per_var = np.array([g.predict(cand)[1] for g in gps])
    mm, mv = _lib.predict_mixture(gps, cand)
    rm, rv = O.mcmc_mixture(per_mu, per_var)
    np.testing.assert_array_equal(mm, rm)
    np.testing.assert_array_equal(mv, rv)
    np.testing.assert_allclose(mm, gold["mix_m"], rtol=MU_RTOL, atol=MU_ATOL)
    np.testing.assert_allclose(mv, gold["mix_v"], rtol=1e-7, atol=1e-10)
    for kind, par in (("ei", 0.0), ("pi", 0.0), ("lcb", 1.0)):
        vals, mx, am, flags = _lib.acq_marginal(gps, kind, par, eta, cand)
        ref = gold["marg_" + kind]
        np.testing.assert_allclose(vals, ref, rtol=ACQ_RTOL, atol=1e-12)
        assert am == int(np.argmax(vals)) and am == int(np.argmax(ref))
    vals, mx, am, _ = _lib.acq_marginal(gps, "log_ei", 0.0, eta, cand)
    # per-sample device LogEI, ordered host accumulation == device accumulation, bit for bit
    per = np.array([g.acq("log_ei", 0.0, eta, cand)[0] for g in gps])
    np.testing.assert_array_equal(vals, per.mean(axis=0))
    z = (eta - gold["mix_m"]) / np.sqrt(gold["mix_v"])
    core = np.abs(z) < 6
    np.testing.assert_allclose(vals[core], gold["marg_log_ei"][core], rtol=1e-5, atol=1e-8)
    # sample-shard partial sums (multi-GPU form) add up to the same thing
    s0, _, _, _ = _lib.acq_marginal(gps[:3], "ei", 0.0, eta, cand, reduce="sum")
    s1, _, _, _ = _lib.acq_marginal(gps[3:], "ei", 0.0, eta, cand, reduce="sum")
    np.testing.assert_allclose((s0 + s1) / len(gps), gold["marg_ei"], rtol=ACQ_RTOL, atol=1e-12)
    cand.close()
    for g in gps:
        g.close()


def check_elementwise(ctx):
    """device element-wise kernel on prescribed moments incl. every degenerate LogEI branch,
    against the reference classes' outputs (tests/golden/acq_elementwise.npz)."""
    gold = load("acq_elementwise")
    m, v, eta = gold["m"], gold["v"], float(gold["eta"])
    z = (eta - m) / np.sqrt(np.where(v > 0, v, 1.0))
    vals, mx, am, flags = _lib.acq_from_moments(ctx, "log_ei", 0.0, eta, m, v)
    assert_logei_close(vals, gold["log_ei"], z, rtol=1e-11, tail_rtol=1e-7)
    vals, _, _, _ = _lib.acq_from_moments(ctx, "log_ei", 0.1, eta, m, v)
    assert_logei_close(vals, gold["log_ei_par"], (eta - 0.1 - m) / np.sqrt(np.where(v > 0, v, 1.0)), rtol=1e-11,
                       tail_rtol=1e-7)
    with np.errstate(all="ignore"):
        vals, _, am, flags = _lib.acq_from_moments(ctx, "pi", 0.0, eta, m, v)
        ok = ~np.isnan(gold["pi"])
        np.testing.assert_allclose(vals[ok], gold["pi"][ok], rtol=1e-11, atol=1e-300)
        assert np.array_equal(np.isnan(vals), np.isnan(gold["pi"]))
        assert am == int(np.argmax(gold["pi"]))          # NaN-first semantics of np.argmax
        assert flags & _lib.FLAG_ZERO_SIGMA
    vals, _, am, _ = _lib.acq_from_moments(ctx, "lcb", 1.0, eta, m, v)
    np.testing.assert_allclose(vals, gold["lcb"], rtol=1e-14)
    assert am == int(np.argmax(gold["lcb"]))
    pos = gold["pos"]
    vals, _, am, flags = _lib.acq_from_moments(ctx, "ei", 0.0, eta, m[pos], v[pos])
    np.testing.assert_allclose(vals, gold["ei_pos"], rtol=1e-10, atol=1e-300)
    assert am == int(np.argmax(gold["ei_pos"])) and not (flags & _lib.FLAG_ZERO_SIGMA)


def check_argmax_semantics(ctx):
    """np.argmax: first maximal index, NaN maximal, -inf everywhere -> 0; independent of geometry."""
    rs = np.random.RandomState(4)
    for n in (1, 63, 64, 65, 255, 256, 257, 1000, 70001):
        m = rs.randn(n)
        v = np.ones(n)
        # LCB with par=0 is -mean: an exact map, so ties survive
        m[rs.randint(n, size=max(1, n // 10))] = -7.0      # many ties at the maximum of -m
        vals, mx, am, _ = _lib.acq_from_moments(ctx, "lcb", 0.0, 0.0, m, v)
        assert am == int(np.argmax(-m)) and mx == 7.0
        if n > 3:
            m2 = m.copy()
            m2[[n // 2, n - 1]] = np.nan
            vals, mx, am, fl = _lib.acq_from_moments(ctx, "lcb", 0.0, 0.0, m2, v)
            assert am == n // 2 and np.isnan(mx) and (fl & _lib.FLAG_NAN)
        vals, mx, am, _ = _lib.acq_from_moments(ctx, "lcb", 0.0, 0.0, np.full(n, np.inf), v)
        assert am == 0 and mx == -np.inf


def check_errors(ctx):
    import pytest
    g = _lib.DeviceGP(ctx, "matern52", 64, 3)
    with pytest.raises(Exception, match="trained first"):
        g.predict(np.zeros((4, 3)))
    rs = np.random.RandomState(0)
    X = rs.rand(20, 3)
    X[3, 0] = np.nan                              # NaN input -> NaN pivot (deterministic failure)
    y = rs.rand(20)
    g.set_data(X, y)
    theta = np.array([0.0, 0.0, 0.0, 0.0, -60.0])
    with pytest.raises(np.linalg.LinAlgError):
        # a NaN pivot is "not positive definite", like LAPACK dpotrf
        g.fit(np.array([10.0, 0.0, 0.0, 0.0, -60.0]), 0.0)
    with pytest.raises(Exception, match="trained first"):
        g.predict(np.zeros((4, 3)))               # failed fit leaves the GP unfitted
    ll, st = g.loglik_batch(np.array([[10.0, 0, 0, 0, -60.0], [0.0, 0, 0, 0, np.log(1e-3)]]), 0.0)
    assert np.all(st == _lib.NOT_POSITIVE_DEFINITE) and np.all(ll == -np.inf)
    with pytest.raises(ValueError):
        g.fit(np.array([np.nan, 0, 0, 0, 0]), 0.0)
    with pytest.raises(AssertionError):
        g.set_data(np.zeros((65, 3)), np.zeros(65))      # n > n_max
    X[3, 0] = 0.5
    g.set_data(X, y)
    g.fit(np.array([0, 0, 0, 0, np.log(1e-2)]), 0.0)
    with pytest.raises(AssertionError):
        g.predict(np.zeros((4, 2)))                       # wrong dim, host array
    with pytest.raises(AssertionError):
        g.predict(_lib.Candidates(ctx, np.zeros((4, 2))))  # wrong dim, resident batch (C ABI: BAD_SHAPE)
    g.close()


def check_edge_sizes(ctx):
    """ragged / tiny / block-boundary sizes: N in {1,2,127,128,129,256}, M in {1,127,128,129}.  N a multiple of 128 (every
    BASELINE size) leaves the augmented row alone in the last block, which is then neither updated nor factored
    (launch_potrf): single-theta fit, batched likelihoods, kept factors and the gradient at such sizes."""
    rs = np.random.RandomState(8)
    D = 3
    theta = np.array([0.3, np.log(0.5), np.log(0.7), np.log(0.9), np.log(1e-3)])
    for N in (1, 2, 127, 128, 129, 256):
        X = rs.rand(N, D)
        y = np.sin(3 * X.sum(axis=1))
        ogp = O.OracleGP("matern52", theta, lower=np.zeros(D), upper=np.ones(D))
        ogp.train(X, y)
        g = _lib.DeviceGP(ctx, "matern52", N, D)
        g.set_data(X, y)
        ll = g.fit(theta, ogp.mean)
        np.testing.assert_allclose(ll, ogp.loglikelihood(theta), rtol=LOGLIK_RTOL, atol=1e-10)
        for M in (1, 127, 128, 129):
            Xc = rs.rand(M, D)
            mu, var = g.predict(Xc)
            mo, vo = ogp.predict(Xc, diag_only=True)
            np.testing.assert_allclose(mu, mo, rtol=MU_RTOL, atol=MU_ATOL)
            np.testing.assert_allclose(var, vo, rtol=0, atol=VAR_ATOL_REL_AMP * np.exp(theta[0]))
        if N % 128 == 0:
            thetas = theta[None, :] + 0.2 * rs.randn(3, theta.size)
            lls, st = g.loglik_batch(thetas, ogp.mean)
            assert np.all(st == _lib.OK)
            np.testing.assert_allclose(lls, [ogp.loglikelihood(t) for t in thetas], rtol=LOGLIK_RTOL, atol=1e-10)
            gps = [_lib.DeviceGP(ctx, "matern52", N, D) for _ in range(3)]
            gps[0].set_data(X, y)
            ll3, st3 = _lib.fit_batch(gps, thetas, ogp.mean)
            assert np.all(st3 == _lib.OK)
            np.testing.assert_array_equal(ll3, lls)
            Xc = rs.rand(64, D)
            for t, gk in zip(thetas, gps):
                o2 = O.OracleGP("matern52", t, lower=np.zeros(D), upper=np.ones(D))
                o2.train(X, y)
                mu, var = gk.predict(Xc)
                mo, vo = o2.predict(Xc, diag_only=True)
                np.testing.assert_allclose(mu, mo, rtol=MU_RTOL, atol=MU_ATOL)
                np.testing.assert_allclose(var, vo, rtol=0, atol=VAR_ATOL_REL_AMP * np.exp(t[0]))
                gk.close()
            llg, grad = g.grad_loglik(theta, ogp.mean)
            np.testing.assert_allclose(llg, ogp.loglikelihood(theta), rtol=LOGLIK_RTOL, atol=1e-10)
            go = O.gp_grad_log_likelihood("matern52", theta, ogp.X, ogp.y, ogp.mean)
            np.testing.assert_allclose(grad, go, rtol=1e-7, atol=1e-8)
        g.close()


def check_uniform_generator(ctx):
    c = _lib.Candidates(ctx, m=5000, dim=7, seed=123)
    P = c.points()
    assert P.shape == (5000, 7) and P.min() >= 0.0 and P.max() < 1.0
    assert abs(P.mean() - 0.5) < 0.01 and abs(P.var() - 1.0 / 12) < 0.005
    assert len(np.unique(P)) == P.size
    c2 = _lib.Candidates(ctx, m=5000, dim=7, seed=123)
    np.testing.assert_array_equal(P, c2.points())
    c3 = _lib.Candidates(ctx, m=5000, dim=7, seed=124)
    assert not np.array_equal(P, c3.points())


def check_fabolas_kernel(ctx):
    """ROBO_KERNEL_FABOLAS (config 4's kernel, robo/fmin/fabolas.py:104-117): gram, fit, posterior
    with the non-stationary prior variance amp (a + b u^2), acquisition."""
    rs = np.random.RandomState(31)
    N, D, M = 150, 4, 300
    s_fid = rs.rand(N) * 0.95 + 0.05
    X = np.concatenate([rs.rand(N, D), ((1 - s_fid) ** 2)[:, None]], axis=1)      # basis (1-s)^2
    y = np.sin(3 * X[:, :D].sum(axis=1)) + 0.5 * X[:, D]
    Xc = np.concatenate([rs.rand(M, D), ((1 - rs.rand(M)) ** 2)[:, None]], axis=1)
    theta = np.concatenate([[0.1], np.log([0.3, 0.5, 0.8, 1.2]), [0.1, -0.3], [np.log(1e-3)]])
    ogp = O.OracleGP("fabolas", theta, normalize_input=False)
    ogp.train(X, y)
    g = _lib.DeviceGP(ctx, "fabolas", N, D + 1)
    assert g.n_theta == theta.size
    g.set_data(X, y)
    K = g.gram(theta)
    Ko = O.kernel_matrix("fabolas", theta[:-1], X) + (ogp.noise + O.JITTER) * np.eye(N)
    np.testing.assert_allclose(K, Ko, rtol=1e-12, atol=1e-14)
    ll = g.fit(theta, ogp.mean)
    np.testing.assert_allclose(ll, ogp.loglikelihood(theta), rtol=LOGLIK_RTOL)
    mu, var = g.predict(Xc)
    mo, vo = ogp.predict(Xc, diag_only=True)
    np.testing.assert_allclose(mu, mo, rtol=MU_RTOL, atol=MU_ATOL)
    np.testing.assert_allclose(var, vo, rtol=0, atol=VAR_ATOL_REL_AMP * Ko.max())
    eta = float(y.min())
    vals, mx, am, _ = g.acq("ei", 0.0, eta, Xc)
    np.testing.assert_allclose(vals, O.ei(mu, var, eta), rtol=1e-10, atol=1e-14)
    assert am == int(np.argmax(O.ei(mo, vo, eta)))
    _, cov = g.predict_cov(Xc[:20])
    covo = O.gp_predict("fabolas", theta, ogp.L, X, y, ogp.mean, Xc[:20])[1]
    np.testing.assert_allclose(cov, covo, rtol=0, atol=VAR_ATOL_REL_AMP * Ko.max())
    g.close()


def check_fp32_gram(ctx):
    """mixed precision of BASELINE config 5: covariance entries in fp32, Cholesky/solves in fp64.
    The device must agree with the oracle's fp32 K-build to fp32 rounding (different libm: a few
    ulp of fp32), and the effect on the posterior vs the all-fp64 path is REPORTED as bounded by
    the conditioning of K, not claimed equal."""
    rs = np.random.RandomState(41)
    N, D, M = 200, 6, 256
    X = rs.rand(N, D)
    y = np.sin(3 * X.sum(axis=1))
    Xc = rs.rand(M, D)
    theta = np.concatenate([[0.0], np.full(D, np.log(0.25 * D)), [np.log(1e-2)]])
    g = _lib.DeviceGP(ctx, "matern52", N, D)
    g.set_data(X, y)
    g.set_precision(True)
    K32 = g.gram(theta)
    K64 = O.kernel_matrix("matern52", theta[:-1], X) + (np.exp(theta[-1]) + O.JITTER) * np.eye(N)
    Ko32 = O.kernel_matrix("matern52", theta[:-1], X, dtype=np.float32) + (np.exp(theta[-1]) + O.JITTER) * np.eye(N)
    assert 1e-9 < np.abs(K32 - K64).max() < 2e-6          # really fp32, and no worse than fp32
    assert np.abs(K32 - Ko32).max() < 1e-6
    o32 = O.OracleGP("matern52", theta, normalize_input=False, dtype=np.float32)
    o32.train(X, y)
    ll = g.fit(theta, o32.mean)
    np.testing.assert_allclose(ll, o32.loglikelihood(theta), rtol=1e-4)
    mu, var = g.predict(Xc)
    mo, vo = o32.predict(Xc, diag_only=True)
    np.testing.assert_allclose(mu, mo, rtol=0, atol=2e-3)
    np.testing.assert_allclose(var, vo, rtol=0, atol=2e-3)
    g.set_precision(False)      # back to fp64: must match the fp64 oracle tightly again
    o64 = O.OracleGP("matern52", theta, normalize_input=False)
    o64.train(X, y)
    g.fit(theta, o64.mean)
    mu, var = g.predict(Xc)
    mo, vo = o64.predict(Xc, diag_only=True)
    np.testing.assert_allclose(mu, mo, rtol=MU_RTOL, atol=MU_ATOL)
    np.testing.assert_allclose(var, vo, rtol=0, atol=VAR_ATOL_REL_AMP)
    g.close()


def check_batched_likelihoods(ctx, sizes=((60, 3), (300, 4))):
    """robo_gp_loglik_batch (one batched pass over S thetas) == S sequential robo_gp_fit calls, bit for
    bit, including a non-finite theta in the middle of the batch; and == the oracle."""
    rs = np.random.RandomState(17)
    for N, D in sizes:
        X = rs.rand(N, D)
        y = np.sin(3 * X.sum(axis=1))
        base = np.concatenate([[0.0], np.full(D, np.log(0.3 * D)), [np.log(1e-2)]])
        S = 7
        thetas = base[None, :] + 0.4 * rs.randn(S, base.size)
        thetas[3, 1] = np.nan
        g = _lib.DeviceGP(ctx, "matern52", N, D)
        g.set_data(X, y)
        mean_c = float(y.mean())
        seq = np.full(S, -np.inf)
        for s in range(S):
            if np.all(np.isfinite(thetas[s])):
                seq[s] = g.fit(thetas[s], mean_c)
        ll, st = g.loglik_batch(thetas, mean_c)
        assert st[3] == _lib.BAD_ARGUMENT and ll[3] == -np.inf
        ok = np.arange(S) != 3
        assert np.all(st[ok] == _lib.OK)
        np.testing.assert_array_equal(ll[ok], seq[ok])
        ogp = O.OracleGP("matern52", thetas[0], normalize_input=False)
        ogp.train(X, y)
        np.testing.assert_allclose(ll[0], ogp.loglikelihood(thetas[0]), rtol=LOGLIK_RTOL)
        import pytest
        with pytest.raises(Exception, match="trained first"):
            g.predict(X[:2])          # the batch call leaves the GP unfitted
        g.close()


def check_batched_split(ctx, N=400, D=3, S=6, variants=((2, 2, -1), (2, 3, 1)), split_min=4):
    """The batched factorisation with its sub-batches on separate streams and staggered group boundaries
    (potrf_split / potrf_group / potrf_lead, potrf.hip launch_potrf): likelihoods AND kept factors bit-identical to the
    one-stream schedule -- every element accumulates the same products in the same order whatever launch carries them."""
    rs = np.random.RandomState(41)
    X = rs.rand(N, D)
    y = np.sin(3 * X.sum(axis=1))
    base = np.concatenate([[0.0], np.full(D, np.log(0.3 * D)), [np.log(1e-2)]])
    thetas = base[None, :] + 0.3 * rs.randn(S, base.size)
    mean_c = float(y.mean())
    g = _lib.DeviceGP(ctx, "matern52", N, D)
    g.set_data(X, y)
    gps = [_lib.DeviceGP(ctx, "matern52", N, D) for _ in range(S)]
    gps[0].set_data(X, y)
    try:
        ctx.set_tuning("potrf_split", 1)
        ctx.set_tuning("potrf_group", 1)
        ctx.set_tuning("potrf_split_min", split_min)
        ref, st = g.loglik_batch(thetas, mean_c)
        assert np.all(st == _lib.OK)
        ll, st = _lib.fit_batch(gps, thetas, mean_c)
        np.testing.assert_array_equal(ll, ref)
        L_ref = [gp.factor().copy() for gp in gps]
        ogp = O.OracleGP("matern52", thetas[1], normalize_input=False)
        ogp.train(X, y)
        np.testing.assert_allclose(ref[1], ogp.loglikelihood(thetas[1]), rtol=LOGLIK_RTOL)
        other = thetas + 0.05
        ctx.set_tuning("potrf_split", 1)
        ref_other, _ = g.loglik_batch(other, mean_c)
        for group, split, lead in variants:
            ctx.set_tuning("potrf_group", group)
            ctx.set_tuning("potrf_split", split)
            ctx.set_tuning("potrf_lead", lead)
            for gram_split in (0, 1):
                ctx.set_tuning("potrf_gram_split", gram_split)
                # NEW thetas first: the workspace must not still hold the gram matrices the side streams are about to read
                # (a side stream that starts before the batch's gram kernel has finished would go unnoticed on repeated inputs)
                ll, st = g.loglik_batch(other, mean_c)
                assert np.all(st == _lib.OK)
                np.testing.assert_array_equal(ll, ref_other, err_msg="group %d split %d lead %d" % (group, split, lead))
                ll, st = g.loglik_batch(thetas, mean_c)
                assert np.all(st == _lib.OK)
                np.testing.assert_array_equal(ll, ref, err_msg="group %d split %d lead %d" % (group, split, lead))
            ctx.set_tuning("potrf_gram_split", None)
            ll, st = _lib.fit_batch(gps, thetas, mean_c)
            np.testing.assert_array_equal(ll, ref)
            for gp, L in zip(gps, L_ref):
                np.testing.assert_array_equal(gp.factor(), L)
    finally:
        for key in ("potrf_split", "potrf_group", "potrf_lead", "potrf_split_min", "potrf_gram_split"):
            ctx.set_tuning(key, None)
        g.close()
        for gp in gps:
            gp.close()


def check_batched_multiple_of_128(ctx, sizes=((256, 3), (384, 2)), S=5, tm4_min=None):
    """N a multiple of 128: the augmented row sits alone in the last block, which is never factored, and its block row is
    updated on 32-row tiles (thin_row) -- inside the 128-row update kernel too (tm4_min lowers the threshold so that the
    interpreter reaches that form at small N).  Batched likelihoods and kept factors == sequential single fits, bit for
    bit, with the thin tiles on and off and on one / three streams."""
    rs = np.random.RandomState(77)
    for N, D in sizes:
        X = rs.rand(N, D)
        y = np.sin(3 * X.sum(axis=1))
        base = np.concatenate([[0.0], np.full(D, np.log(0.3 * D)), [np.log(1e-2)]])
        thetas = base[None, :] + 0.3 * rs.randn(S, base.size)
        mean_c = float(y.mean())
        g = _lib.DeviceGP(ctx, "matern52", N, D)
        g.set_data(X, y)
        seq, L_seq = [], []
        for th in thetas:
            seq.append(g.fit(th, mean_c))
            L_seq.append(g.factor().copy())
        gps = [_lib.DeviceGP(ctx, "matern52", N, D) for _ in range(S)]
        gps[0].set_data(X, y)
        try:
            if tm4_min is not None:
                ctx.set_tuning("potrf_batch_tm4_min", tm4_min)
            for thin, split in ((1, 1), (0, 1), (1, 3)):
                ctx.set_tuning("potrf_thin_last", thin)
                ctx.set_tuning("potrf_split", split)
                ctx.set_tuning("potrf_split_min", 2)
                ll, st = g.loglik_batch(thetas, mean_c)
                assert np.all(st == _lib.OK)
                np.testing.assert_array_equal(ll, np.array(seq), err_msg="N %d thin %d split %d" % (N, thin, split))
                ll, st = _lib.fit_batch(gps, thetas, mean_c)
                np.testing.assert_array_equal(ll, np.array(seq))
                for gp, L in zip(gps, L_seq):
                    np.testing.assert_array_equal(gp.factor(), L)
        finally:
            for key in ("potrf_batch_tm4_min", "potrf_thin_last", "potrf_split", "potrf_split_min"):
                ctx.set_tuning(key, None)
            g.close()
            for gp in gps:
                gp.close()


def check_fit_batch(ctx, sizes=((60, 3), (300, 4)), kind="matern52"):
    """robo_gp_fit_batch (the per-sample model fits of GaussianProcessMCMC.train in one batched pass that keeps
    the factors) == S sequential robo_gp_fit calls on S handles: log-likelihood, Cholesky factor and posterior bit
    for bit; a sample with a non-PD K is reported and left unfitted without disturbing its neighbours."""
    import pytest
    rs = np.random.RandomState(29)
    for N, D in sizes:
        X = rs.rand(N, D)
        y = np.sin(3 * X.sum(axis=1))
        P = O.n_kernel_params(kind, D) + 1
        base = np.zeros(P)
        base[1:1 + D if kind != "fabolas" else D] = np.log(0.3 * D)
        base[-1] = np.log(1e-2)
        S = 6
        thetas = base[None, :] + 0.4 * rs.randn(S, P)
        thetas[2, 0] = 710.0                          # amplitude e^710 = inf: inf / inf = NaN pivots (deterministic)
        mean_c = float(y.mean())
        Xc = rs.rand(40, D)
        seq = []
        for s in range(S):
            g = _lib.DeviceGP(ctx, kind, N, D)
            g.set_data(X, y)
            try:
                ll = g.fit(thetas[s], mean_c)
                seq.append((ll, g.factor(), g.predict(Xc)))
            except np.linalg.LinAlgError:
                seq.append(None)
            g.close()
        assert seq[2] is None, "the stress sample was meant to be not positive definite"
        gps = [_lib.DeviceGP(ctx, kind, N + 7 * s, D) for s in range(S)]     # different capacities
        gps[0].set_data(X, y)
        ll, st = _lib.fit_batch(gps, thetas, mean_c)
        for s in range(S):
            if seq[s] is None:
                assert st[s] == _lib.NOT_POSITIVE_DEFINITE and ll[s] == -np.inf
                with pytest.raises(Exception, match="trained first"):
                    gps[s].predict(Xc)
                continue
            assert st[s] == _lib.OK
            assert ll[s] == seq[s][0]
            np.testing.assert_array_equal(gps[s].factor(), seq[s][1])
            mu, var = gps[s].predict(Xc)
            np.testing.assert_array_equal(mu, seq[s][2][0])
            np.testing.assert_array_equal(var, seq[s][2][1])
        # a kept handle is a full handle: it can be refitted on its own copy of the data
        ll1 = gps[1].fit(thetas[0], mean_c)
        assert ll1 == seq[0][0]
        for g in gps:
            g.close()


def check_grad_loglik(ctx, cases=(("matern52", 70, 3), ("rbf", 200, 5), ("fabolas", 150, 4), ("matern52", 300, 20),
                                  ("matern52", 650, 2))):
    """robo_gp_grad_loglik == the oracle's restatement of GaussianProcess.grad_nll
    (robo/models/gaussian_process.py:168-191) for every kernel kind, several block counts (N below,
    across and well above one 128 block) and D above one 16-dimension staging pass; the GP is left
    fitted at theta; the call is deterministic (bitwise repeatable)."""
    rs = np.random.RandomState(23)
    for kind, N, D in cases:
        X = rs.rand(N, D)
        y = np.sin(3 * X.sum(axis=1)) + 0.1 * rs.randn(N)
        P = O.n_kernel_params(kind, D) + 1
        theta = 0.3 * rs.randn(P)
        theta[1:P - 1] += np.log(0.3 * D) if kind != "fabolas" else 0.0
        theta[-1] = np.log(1e-2)
        mean_c = float(y.mean())
        g = _lib.DeviceGP(ctx, kind, N, D)
        g.set_data(X, y)
        ll, grad = g.grad_loglik(theta, mean_c)
        ref = O.gp_grad_log_likelihood(kind, theta, X, y, mean_c)
        L = O.gp_compute(kind, theta, X)
        np.testing.assert_allclose(ll, O.gp_log_likelihood(L, y, mean_c), rtol=LOGLIK_RTOL)
        scale = np.max(np.abs(ref))
        np.testing.assert_allclose(grad, ref, rtol=1e-8, atol=1e-9 * scale, err_msg="%s N=%d D=%d" % (kind, N, D))
        ll2, grad2 = g.grad_loglik(theta, mean_c)
        assert ll2 == ll
        np.testing.assert_array_equal(grad2, grad)
        mu, var = g.predict(X[:5])          # fitted at theta afterwards
        assert np.all(np.isfinite(mu)) and np.all(var > 0)
        g.close()


def check_model_gradients(ctx):
    """GaussianProcess.grad_nll (host mirror) == the oracle restatement of the reference's grad_nll
    incl. the prior term and the noise quirk; the gradient-based optimize() (use_gradients=True,
    BFGS on the consistent gradient) ends at an nll no worse than the finite-difference L-BFGS-B
    default, from the same start."""
    from robo_amd.kernels import Matern52Kernel
    from robo_amd.models import GaussianProcess
    from robo_amd.priors import DefaultPrior
    rs = np.random.RandomState(5)
    N, D = 90, 3
    X = rs.rand(N, D)
    y = np.sinc(X * 10 - 5).sum(axis=1)
    lower, upper = np.zeros(D), np.ones(D)
    models = {}
    for ug in (False, True):
        kernel = 2.0 * Matern52Kernel(np.ones(D), ndim=D)
        prior = DefaultPrior(len(kernel) + 1, rng=np.random.RandomState(0))
        m = GaussianProcess(kernel, prior=prior, use_gradients=ug, lower=lower, upper=upper,
                            rng=np.random.RandomState(1))
        m.train(X, y, do_optimize=True)
        models[ug] = m
    m = models[True]
    theta = m.hypers + 0.05
    ref = -(O.gp_grad_log_likelihood("matern52", theta, m.X, m.y, m.mean) + m.prior.gradient(theta))
    np.testing.assert_allclose(m.grad_nll(theta), ref, rtol=1e-7, atol=1e-8 * np.max(np.abs(ref)))
    # consistent objective/gradient pair: central differences of the objective itself
    f0, g0 = m._nll_with_gradient(theta)
    for p in range(theta.size):
        e = np.zeros_like(theta)
        e[p] = 1e-5
        fd = (m._nll_with_gradient(theta + e)[0] - m._nll_with_gradient(theta - e)[0]) / 2e-5
        assert abs(fd - g0[p]) <= 1e-4 * max(1.0, abs(fd)), (p, fd, g0[p])
    assert models[True].nll(models[True].hypers) <= models[False].nll(models[False].hypers) + 1e-3 * abs(
        models[False].nll(models[False].hypers))
    mu, var = models[True].predict(X[:4])
    assert np.all(np.isfinite(mu)) and np.all(var > 0)


def check_ill_conditioned(ctx, cases=((300, 1, 1e-8, 0.3), (260, 2, 1e-6, 0.5))):
    """Dense designs with tiny noise (cond(K) 1e8 .. 1e11 at the GPU sizes): the blocked factorisation
    with explicit inverses of its diagonal blocks stays within a few ulp * cond of the oracle's LAPACK
    substitution path (measured on the MI355X at cond 8e10: |d mu| 2e-11, |d var| 2e-13, rel d loglik
    1e-10)."""
    rs = np.random.RandomState(0)
    for N, D, noise, ls in cases:
        X = rs.rand(N, D)
        y = np.sin(4 * X.sum(axis=1))
        theta = np.concatenate([[0.0], np.full(D, np.log(ls ** 2)), [np.log(noise)]])
        g = _lib.DeviceGP(ctx, "matern52", N, D)
        g.set_data(X, y)
        c = float(y.mean())
        ll = g.fit(theta, c)
        L = O.gp_compute("matern52", theta, X)
        np.testing.assert_allclose(ll, O.gp_log_likelihood(L, y, c), rtol=2e-9)
        Xs = rs.rand(500, D)
        mu, var = g.predict(Xs)
        muo, varo = O.gp_predict_diag("matern52", theta, L, X, y, c, Xs)
        np.testing.assert_allclose(mu, muo, rtol=0, atol=1e-9)
        np.testing.assert_allclose(var, varo, rtol=0, atol=1e-11)
        g.close()


def check_device_random_candidates(ctx):
    """device-generated RandomSampling recipe: bounds, split, moments; winner row read-back"""
    loc = np.array([0.2, 0.9, 0.5])
    scale = np.array([0.1, 0.05, 0.02])
    m, nu = 20000, 14000
    c = _lib.Candidates(ctx, m=m, seed=5, n_uniform=nu, loc=loc, scale=scale)
    P_ = c.points()
    assert P_.shape == (m, 3) and P_.min() >= 0.0 and P_.max() <= 1.0
    U, G = P_[:nu], P_[nu:]
    assert abs(U.mean() - 0.5) < 0.01 and abs(U.var() - 1 / 12.0) < 0.005
    assert np.all(np.abs(G[:, 2].mean() - 0.5) < 0.002) and abs(G[:, 2].std() - 0.02) < 0.002
    assert abs(G[:, 0].mean() - 0.2) < 0.01                   # mildly clipped at 0
    assert (G[:, 1] == 1.0).mean() > 0.01                     # clipping at the upper bound happens
    np.testing.assert_array_equal(c.point(nu + 7), P_[nu + 7])
    np.testing.assert_array_equal(_lib.Candidates(ctx, m=m, seed=5, n_uniform=nu, loc=loc, scale=scale).points(), P_)


def check_candidate_reupload(ctx):
    """robo_cand_set_points: a new batch of the same shape into an existing handle (the per-iteration H2D of a BO
    loop) == a fresh handle, bit for bit; wrong shapes are refused"""
    import pytest
    rs = np.random.RandomState(77)
    N, D, M = 150, 4, 333
    X = rs.rand(N, D)
    y = np.sin(3 * X.sum(axis=1))
    theta = np.concatenate([[0.1], np.log(0.3 * D) + 0.1 * rs.randn(D), [np.log(1e-3)]])
    g = _lib.DeviceGP(ctx, "matern52", N, D)
    g.set_data(X, y)
    g.fit(theta, float(y.mean()))
    A, B = rs.rand(M, D), rs.rand(M, D)
    c = _lib.Candidates(ctx, A)
    first = g.acq("ei", 0.0, float(y.min()), c)
    c.set_points(B)
    again = g.acq("ei", 0.0, float(y.min()), c)
    fresh = _lib.Candidates(ctx, B)
    want = g.acq("ei", 0.0, float(y.min()), fresh)
    np.testing.assert_array_equal(again[0], want[0])
    assert again[1:] == want[1:] and not np.array_equal(first[0], again[0])
    np.testing.assert_array_equal(c.points(), B)
    with pytest.raises(AssertionError):
        c.set_points(rs.rand(M + 1, D))
    for h in (c, fresh, g):
        h.close()


def check_shape_sweep(ctx, n_cases=40, seed=123, max_n=700):
    """randomised shapes: N, D (incl. D > 16: several LDS coordinate chunks), M, kernel kind, output
    normalisation; fit + posterior + EI argmax against the oracle."""
    rs = np.random.RandomState(seed)
    dims = [1, 2, 3, 5, 8, 15, 16, 17, 31, 33, 64, 100]
    for case in range(n_cases):
        N = int(rs.choice([1, 2, 3, 17, 100, 127, 128, 129, 255, 256, 257, 300, 511, max_n]))
        D = int(rs.choice(dims))
        M = int(rs.choice([1, 5, 64, 127, 128, 129, 300, 700]))
        kind = str(rs.choice(["matern52", "rbf", "fabolas"])) if D >= 2 else str(rs.choice(["matern52", "rbf"]))
        nout = bool(rs.rand() < 0.3) and N > 1
        X = rs.rand(N, D)
        y = np.sin(3 * X.sum(axis=1) / np.sqrt(D)) + 0.05 * rs.randn(N)
        Xc = rs.rand(M, D)
        if kind == "fabolas":
            theta = np.concatenate([[0.1], np.log(0.3 + rs.rand(D - 1)), [0.1, -0.3], [np.log(1e-2)]])
        else:
            theta = np.concatenate([[0.2], np.log((0.2 + rs.rand(D)) * D), [np.log(1e-2)]])
        ogp = O.OracleGP(kind, theta, normalize_input=False, normalize_output=nout)
        ogp.train(X, y)
        g = _lib.DeviceGP(ctx, kind, N, D)
        g.set_data(ogp.X, ogp.y)
        if nout:
            g.set_output_transform(ogp.y_mean, ogp.y_std)
        tag = "case %d: N=%d D=%d M=%d %s nout=%s" % (case, N, D, M, kind, nout)
        ll = g.fit(theta, ogp.mean)
        np.testing.assert_allclose(ll, ogp.loglikelihood(theta), rtol=LOGLIK_RTOL, atol=1e-9, err_msg=tag)
        mu, var = g.predict(Xc)
        mo, vo = ogp.predict(Xc, diag_only=True)
        scale = max(1.0, np.abs(mo).max())
        amp = O.kernel_diag(kind, theta[:-1], Xc).max() * (ogp.y_std ** 2 if nout else 1.0)
        np.testing.assert_allclose(mu, mo, rtol=MU_RTOL, atol=MU_ATOL * scale, err_msg=tag)
        np.testing.assert_allclose(var, vo, rtol=0, atol=VAR_ATOL_REL_AMP * max(amp, 1.0), err_msg=tag)
        _, eta = ogp.get_incumbent()
        vals, mx, am, _ = g.acq("ei", 0.0, float(eta), Xc)
        eo = O.ei(mo, vo, eta)
        want = int(np.argmax(eo))
        srt = np.sort(eo)
        gap = srt[-1] - srt[-2] if M > 1 else 1.0
        assert am == want or gap <= 1e-7 * max(abs(eo[want]), 1e-300), tag
        g.close()


def check_predictive_gradients(ctx, cases=(("matern52", 70, 3, 9), ("rbf", 150, 5, 40), ("fabolas", 140, 4, 33)),
                               device=None):
    """robo_gp_predict_grad (D + 1 right-hand sides of the blocked forward substitution per point) and the
    host classes' predictive_gradients / derivative=True against the oracle's analytic gradients (themselves pinned
    to central differences in tests/test_oracle.py) and against central differences of the DEVICE's own predict."""
    from robo_amd import acquisition_functions as A
    from robo_amd.kernels import ExpSquaredKernel, FabolasKernel, Matern52Kernel
    from robo_amd.models import FabolasGP, GaussianProcess
    rs = np.random.RandomState(41)
    for kind, N, D, M in cases:
        lower, upper = np.full(D, -1.0), np.full(D, 2.0)
        X01 = rs.rand(N, D)
        X = lower + (upper - lower) * X01
        y = np.sin(3 * X01.sum(axis=1)) * 1.5 + 0.2
        P = O.n_kernel_params(kind, D) + 1
        theta = 0.3 * rs.randn(P)
        theta[-1] = np.log(1e-2)
        Xt01 = rs.rand(M, D)
        # ---- C ABI on the GP's own input space
        ogp = O.OracleGP(kind, theta, normalize_input=False)
        ogp.train(X01, y)
        g = _lib.DeviceGP(ctx, kind, N, D)
        g.set_data(X01, y)
        g.fit(theta, ogp.mean)
        mean, var, dm, dv = g.predict_grad(Xt01)
        mo, vo = ogp.predict(Xt01, diag_only=True)
        dmo, dvo = ogp.predictive_gradients(Xt01)
        amp = float(np.max(O.kernel_diag(kind, theta[:-1], Xt01)))
        np.testing.assert_allclose(mean, mo, rtol=MU_RTOL, atol=MU_ATOL)
        np.testing.assert_allclose(var, vo, rtol=0, atol=VAR_ATOL_REL_AMP * amp)
        np.testing.assert_allclose(dm, dmo[:, :, 0], rtol=1e-8, atol=1e-9 * np.abs(dmo).max())
        np.testing.assert_allclose(dv, dvo, rtol=1e-7, atol=1e-8 * max(np.abs(dvo).max(), amp))
        # the same values as the plain posterior path (fused kernel)
        m2, v2 = g.predict(Xt01)
        np.testing.assert_allclose(mean, m2, rtol=1e-11, atol=1e-12)
        np.testing.assert_allclose(var, v2, rtol=0, atol=1e-11 * amp)
        # central differences of the device's own posterior
        h = 1e-5
        for d in (0, D - 1):
            e = np.zeros(D)
            e[d] = h
            mp, vp = g.predict(Xt01 + e)
            mm, vm = g.predict(Xt01 - e)
            np.testing.assert_allclose(dm[:, d], (mp - mm) / (2 * h), rtol=1e-5, atol=1e-6)
            np.testing.assert_allclose(dv[:, d], (vp - vm) / (2 * h), rtol=1e-4, atol=1e-6 * amp)
        g.close()
        # ---- host classes: input / output normalisation chain rule, GPy shapes, acquisition derivatives
        if kind == "fabolas":
            kern = FabolasKernel(D)
            kern.set_parameter_vector(theta[:-1])
            model = FabolasGP(kern, basis_function=lambda s: (1 - s) ** 2, noise=np.exp(theta[-1]), lower=lower[:-1],
                              upper=upper[:-1], rng=np.random.RandomState(1), device=device)
            Xr = np.concatenate((X[:, :-1], X01[:, -1:]), axis=1)
            Xq = np.concatenate((lower[:-1] + (upper[:-1] - lower[:-1]) * Xt01[:, :-1], Xt01[:, -1:]), axis=1)
        else:
            kern = 1.0 * (Matern52Kernel if kind == "matern52" else ExpSquaredKernel)(np.ones(D), ndim=D)
            kern.set_parameter_vector(theta[:-1])
            model = GaussianProcess(kern, noise=np.exp(theta[-1]), normalize_output=(kind == "rbf"), lower=lower,
                                    upper=upper, rng=np.random.RandomState(1), device=device)
            Xr, Xq = X, lower + (upper - lower) * Xt01
        model.train(Xr, y, do_optimize=False)
        dmdx, dvdx = model.predictive_gradients(Xq)
        assert dmdx.shape == (M, D, 1) and dvdx.shape == (M, D)
        h = 1e-5
        for d in range(D):
            e = np.zeros(D)
            e[d] = h
            mp, vp = model.predict(Xq + e)
            mm, vm = model.predict(Xq - e)
            np.testing.assert_allclose(dmdx[:, d, 0], (mp - mm) / (2 * h), rtol=1e-5, atol=1e-6)
            np.testing.assert_allclose(dvdx[:, d], (vp - vm) / (2 * h), rtol=1e-4, atol=1e-6)
        for cls in (A.EI, A.PI, A.LCB):
            acq = cls(model)
            f, df = acq.compute(Xq, derivative=True)
            assert df.shape == (M, D)
            np.testing.assert_allclose(f, acq.compute(Xq), rtol=1e-12)
            for d in (0, D - 1):
                e = np.zeros(D)
                e[d] = h
                fd = (acq.compute(Xq + e) - acq.compute(Xq - e)) / (2 * h)
                np.testing.assert_allclose(df[:, d], fd, rtol=2e-4, atol=1e-6 * max(1.0, np.abs(fd).max()))
        # the single-point form the reference's optimisers use (1, D) -> (1, D)
        f1, df1 = A.EI(model).compute(Xq[:1], derivative=True)
        assert df1.shape == (1, D)
        if kind == "matern52" and N <= 200:
            # robo/util/posterior_optimization.py with_gradients=True: the consumer of predictive_gradients
            from robo_amd.util.posterior_optimization import posterior_mean_optimization, \
                posterior_mean_plus_std_optimization
            np.random.seed(3)
            xa = posterior_mean_optimization(model, lower, upper, n_restarts=3, with_gradients=True)
            np.random.seed(3)
            xb = posterior_mean_optimization(model, lower, upper, n_restarts=3, with_gradients=False)
            fa, fb = model.predict(xa[None, :])[0][0], model.predict(xb[None, :])[0][0]
            assert fa <= fb + 1e-6 * max(1.0, abs(fb))          # analytic gradients do at least as well
            np.random.seed(3)
            xc = posterior_mean_plus_std_optimization(model, lower, upper, n_restarts=2, with_gradients=True)
            assert np.all(xc >= lower) and np.all(xc <= upper)
        if getattr(model, "gp", None) is not None:
            model.gp.close()


def check_sobol_candidates(ctx, dims=(3, 64), m=1000):
    """robo_cand_create_sobol == scipy.stats.qmc.Sobol(scramble=True) bit for bit, any slice of the sequence
    (the per-rank slices of BASELINE config 5's candidate shard), and the unscrambled sequence too"""
    from scipy.stats import qmc
    for d in dims:
        for scramble in (True, False):
            eng = qmc.Sobol(d=d, scramble=scramble, seed=0)
            c = _lib.Candidates(ctx, m=m, sobol=eng, first=0)
            got = c.points()
            c.close()
            ref = qmc.Sobol(d=d, scramble=scramble, seed=0)
            import warnings
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                want = ref.random(m)
            np.testing.assert_array_equal(got, want)
            assert got.min() >= 0.0 and got.max() < 1.0
            # a later slice (rank 3 of 8 with 300 points per rank)
            c = _lib.Candidates(ctx, m=300, sobol=eng, first=900)
            ref = qmc.Sobol(d=d, scramble=scramble, seed=0)
            ref.fast_forward(900)
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                want = ref.random(300)
            np.testing.assert_array_equal(c.points(), want)
            c.close()


def check_phase_events(ctx):
    """robo_ctx_set_phase_events: the library brackets the phases of robo_gp_fit with event slots 19..23 only when
    asked to (off by default: the event packets cost a 1.8 ms fit ~30 us); caller slots work either way."""
    rs = np.random.RandomState(3)
    N, D = 200, 3
    X = rs.rand(N, D)
    y = np.sin(3 * X.sum(axis=1))
    theta = np.concatenate([[0.0], np.full(D, np.log(0.5)), [np.log(1e-2)]])
    g = _lib.DeviceGP(ctx, "matern52", N, D)
    g.set_data(X, y)
    ll0 = g.fit(theta, 0.0)
    ctx.set_phase_events(True)
    ll1 = g.fit(theta, 0.0)
    gram, chol, llk, k1 = ctx.elapsed_ms(20, 21), ctx.elapsed_ms(21, 22), ctx.elapsed_ms(22, 23), ctx.elapsed_ms(19, 21)
    ctx.set_phase_events(False)
    assert ll1 == ll0
    for v in (gram, chol, llk, k1):
        assert np.isfinite(v) and v >= 0.0
    assert chol > 0.0 and k1 <= gram + 1e-3
    ctx.record(0)
    ctx.record(1)
    assert ctx.elapsed_ms(0, 1) >= 0.0
    g.close()


def check_small_and_large_tiles_agree(ctx, monkeypatch, cases=(("matern52", 300, 5, 700), ("fabolas", 200, 4, 130))):
    """the 32-candidate block-row step (small batches) and the 128-candidate one give the same bits"""
    rs = np.random.RandomState(41)
    for kind, N, D, M in cases:
        X = rs.rand(N, D)
        y = np.sin(3 * X.sum(axis=1))
        P = O.n_kernel_params(kind, D) + 1
        theta = np.zeros(P)
        theta[1:1 + D if kind != "fabolas" else D] = np.log(0.4 * D)
        theta[-1] = np.log(1e-2)
        g = _lib.DeviceGP(ctx, kind, N, D)
        g.set_data(X, y)
        g.fit(theta, float(y.mean()))
        Xc = rs.rand(M, D)
        try:
            ctx.set_tuning("winv_max", 0)           # the block-row substitution only (not the explicit-inverse path)
            ctx.set_tuning("trsm_small_max", 0)
            ctx.set_tuning("trsm_pair", 0)          # one block row per launch
            cl = _lib.Candidates(ctx, Xc)
            mu_l, var_l = g.predict(cl)
            assert cl.solve_kernel() == "trsm_step_gen_kernel"
            _, mx_l, am_l, _ = g.acq("ei", 0.0, float(y.min()), Xc)
            ctx.set_tuning("trsm_pair", 1)          # two block rows per launch on one read of V: same bits
            mu_p, var_p = g.predict(cl)
            assert cl.solve_kernel() == "trsm_pair_gen_kernel"
            _, mx_p, am_p, _ = g.acq("ei", 0.0, float(y.min()), Xc)
            cl.close()
            np.testing.assert_array_equal(mu_p, mu_l)
            np.testing.assert_array_equal(var_p, var_l)
            assert (mx_p, am_p) == (mx_l, am_l)
            ctx.set_tuning("trsm_small_max", 1000000)
            for narrow in (0, 1):          # 32 and 16 candidates per workgroup
                for deep in (0, 1):        # one and two k-tiles per staging stage
                    ctx.set_tuning("trsm_small_narrow", narrow)
                    ctx.set_tuning("trsm_small_deep", deep)
                    mu_s, var_s = g.predict(Xc)
                    _, mx_s, am_s, _ = g.acq("ei", 0.0, float(y.min()), Xc)
                    np.testing.assert_array_equal(mu_s, mu_l)
                    np.testing.assert_array_equal(var_s, var_l)
                    assert (mx_s, am_s) == (mx_l, am_l)
        finally:
            for key in ("trsm_small_narrow", "trsm_small_deep", "trsm_small_max", "winv_max", "trsm_pair"):
                ctx.set_tuning(key, None)
        g.close()


def check_host_array_handle_reuse(ctx):
    """the host-array entry points keep their candidate handle inside the GP between calls of one batch size: results
    must not depend on what the handle held before (other sizes, other points, a refit in between)"""
    rs = np.random.RandomState(17)
    N, D = 150, 4
    X = rs.rand(N, D)
    y = np.sin(3 * X.sum(axis=1))
    theta = np.concatenate([[0.0], np.full(D, np.log(0.5)), [np.log(1e-2)]])
    theta2 = theta.copy()
    theta2[1:1 + D] += 0.4
    g = _lib.DeviceGP(ctx, "matern52", N, D)
    g.set_data(X, y)
    g.fit(theta, 0.0)

    def fresh(th, Xc):
        h = _lib.DeviceGP(ctx, "matern52", N, D)
        h.set_data(X, y)
        h.fit(th, 0.0)
        out = h.predict(Xc), h.acq("ei", 0.0, float(y.min()), Xc)
        h.close()
        return out

    A, B, C_ = rs.rand(50, D), rs.rand(70, D), rs.rand(50, D)
    for th, Xc in ((theta, A), (theta, B), (theta, C_), (theta2, C_), (theta2, A), (theta, rs.rand(1, D))):
        g.fit(th, 0.0)
        (mu_f, var_f), (val_f, mx_f, am_f, fl_f) = fresh(th, Xc)
        mu, var = g.predict(Xc)
        val, mx, am, fl = g.acq("ei", 0.0, float(y.min()), Xc)
        np.testing.assert_array_equal(mu, mu_f)
        np.testing.assert_array_equal(var, var_f)
        np.testing.assert_array_equal(val, val_f)
        assert (mx, am, fl) == (mx_f, am_f, fl_f)
    g.close()



def check_chunked_workspace(ctx, monkeypatch, N, D, M, ws_blocks, kind="matern52"):
    """Candidate batches larger than the solve workspace (tuning key ws_bytes) are evaluated in passes of whole
    128-candidate blocks: posterior, acquisition values and argmax must equal the single-pass results bit for bit,
    and the oracle's within the stated tolerances."""
    from _tol import MU_ATOL, MU_RTOL, VAR_ATOL_REL_AMP
    rs = np.random.RandomState(N + M)
    X = rs.rand(N, D)
    y = np.sinc(X * 10 - 5).sum(axis=1)
    theta = np.concatenate([[0.2], np.log(0.25 * D) + 0.2 * rs.randn(D), [np.log(1e-3)]])
    Xc = rs.rand(M, D)
    ogp = O.OracleGP(kind, theta, lower=np.zeros(D), upper=np.ones(D))
    ogp.train(X, y)
    g = _lib.DeviceGP(ctx, kind, N, D)
    g.set_data(X, y)
    g.fit(theta, ogp.mean)
    eta = float(y.min())
    ctx.set_tuning("ws_bytes", None)
    mu1, var1 = g.predict(Xc)
    v1, mx1, am1, _ = g.acq("ei", 0.0, eta, Xc)
    n_pad = (N + 1 + 127) // 128 * 128
    ctx.set_tuning("ws_bytes", ws_blocks * 128 * n_pad * 8)
    cand = _lib.Candidates(ctx, Xc)                 # a fresh handle: its workspace is sized under the limit
    mu2, var2 = g.predict(cand)
    v2, mx2, am2, _ = g.acq("ei", 0.0, eta, cand)
    n_pass = -(-((M + 127) // 128) // ws_blocks)
    assert n_pass >= 2, n_pass
    assert cand.chunk() == ws_blocks * 128, (cand.chunk(), ws_blocks)
    cand.close()
    ctx.set_tuning("ws_bytes", None)
    np.testing.assert_array_equal(mu1, mu2)
    np.testing.assert_array_equal(var1, var2)
    np.testing.assert_array_equal(v1, v2)
    assert am1 == am2 and mx1 == mx2
    mu_o, var_o = ogp.predict(Xc, diag_only=True)
    np.testing.assert_allclose(mu2, mu_o, rtol=MU_RTOL, atol=MU_ATOL)
    np.testing.assert_allclose(var2, var_o, rtol=0, atol=VAR_ATOL_REL_AMP * np.exp(theta[0]))
    ei_o = O.ei(mu_o, var_o, eta)
    want, srt = int(np.argmax(ei_o)), np.sort(ei_o)
    assert am2 == want or srt[-1] - srt[-2] <= 1e-7 * abs(ei_o[want]), (am2, want)
    g.close()
    return n_pass


def check_winv_path(ctx, cases=(("matern52", 300, 5, 700), ("fabolas", 280, 4, 130), ("rbf", 1100, 3, 40))):
    """Small batches through the explicit inverse factor W = L^-1 (winv.hip) against the oracle at the stated
    tolerances and against the block-row substitution (rounding-level agreement, same argmax); values independent of
    the workspace chunking (bit for bit); full covariance and cross-covariances (callers that consume V itself);
    a refit rebuilds W; an ill-conditioned factor stays on the substitution."""
    from _tol import MU_ATOL, MU_RTOL, VAR_ATOL_REL_AMP
    rs = np.random.RandomState(43)
    try:
        ctx.set_tuning("winv_min_blocks", 2)
        import itertools
        # both forms of the product: chunked units + reduction (small batches) and one workgroup per (tile, block row) over
        # the whole contraction range with the reductions in its epilogue (batches that fill the chip) -- forced in turn
        for (kind, N, D, M), rows_mode in itertools.product(cases, (0, 1)):
            ctx.set_tuning("winv_rows", rows_mode)
            winv_name = "winv_row_kernel" if rows_mode else "winv_gemm_kernel"
            X = rs.rand(N, D)
            y = np.sin(3 * X.sum(axis=1))
            P = O.n_kernel_params(kind, D) + 1
            theta = np.zeros(P)
            theta[1:1 + D if kind != "fabolas" else D] = np.log(0.4 * D)
            theta[-1] = np.log(1e-2)
            ogp = O.OracleGP(kind, theta, normalize_input=False)
            ogp.train(X, y)
            g = _lib.DeviceGP(ctx, kind, N, D)
            g.set_data(X, y)
            g.fit(theta, ogp.mean)
            Xc = rs.rand(M, D)
            eta = float(y.min())
            amp = float(np.max(O.kernel_diag(kind, theta[:-1], Xc)))
            ctx.set_tuning("winv_max", 0)
            mu_b, var_b = g.predict(Xc)
            _, _, am_b, _ = g.acq("ei", 0.0, eta, Xc)
            ctx.set_tuning("winv_max", None)
            cand = _lib.Candidates(ctx, Xc)
            mu_w, var_w = g.predict(cand)
            assert cand.solve_kernel() == winv_name
            vals, mx, am_w, _ = g.acq("ei", 0.0, eta, cand)
            # the other form of the product on the same batch: rounding-level agreement
            ctx.set_tuning("winv_rows", 1 - rows_mode)
            mu_x, var_x = g.predict(cand)
            assert cand.solve_kernel() != winv_name and cand.solve_kernel().startswith("winv_")
            np.testing.assert_allclose(mu_x, mu_w, rtol=0, atol=3e-11 * max(1.0, np.abs(mu_w).max()))
            np.testing.assert_allclose(var_x, var_w, rtol=0, atol=3e-11 * amp)
            ctx.set_tuning("winv_rows", rows_mode)
            mu_o, var_o = ogp.predict(Xc, diag_only=True)
            np.testing.assert_allclose(mu_w, mu_o, rtol=MU_RTOL, atol=MU_ATOL)
            np.testing.assert_allclose(var_w, var_o, rtol=0, atol=VAR_ATOL_REL_AMP * amp)
            np.testing.assert_allclose(mu_w, mu_b, rtol=0, atol=1e-11 * max(1.0, np.abs(mu_b).max()))
            np.testing.assert_allclose(var_w, var_b, rtol=0, atol=1e-11 * amp)
            if not rows_mode:
                # the chunked form's three unit depths (by default picked from the batch size: whole batches, 3..7 candidate
                # tiles, 1..2): each within the stated tolerances of the oracle, rounding-level agreement with each other
                for shift in (0, 1, 2):
                    ctx.set_tuning("winv_kc_shift", shift)
                    mu_k, var_k = g.predict(cand)
                    assert cand.solve_kernel() == "winv_gemm_kernel"
                    np.testing.assert_allclose(mu_k, mu_o, rtol=MU_RTOL, atol=MU_ATOL)
                    np.testing.assert_allclose(var_k, var_o, rtol=0, atol=VAR_ATOL_REL_AMP * amp)
                    np.testing.assert_allclose(mu_k, mu_w, rtol=0, atol=3e-11 * max(1.0, np.abs(mu_w).max()))
                    np.testing.assert_allclose(var_k, var_w, rtol=0, atol=3e-11 * amp)
                ctx.set_tuning("winv_kc_shift", None)
            ei_o = O.ei(mu_o, var_o, eta)
            want, srt = int(np.argmax(ei_o)), np.sort(ei_o)
            assert am_w == int(np.argmax(vals)) and (am_w == want or srt[-1] - srt[-2] <= 1e-7 * abs(ei_o[want]))
            assert am_w == am_b or srt[-1] - srt[-2] <= 1e-7 * abs(ei_o[want])
            if not rows_mode:
                # a handful of candidates (the reference's 1 x D callers): the matrix-vector form, against the oracle and
                # against the chunked form of the same points; full covariance through it (a consumer of V itself)
                ctx.set_tuning("winv_rows", None)
                for m_few in (1, 5, 8):
                    few = _lib.Candidates(ctx, Xc[:m_few])
                    mu_f, var_f = g.predict(few)
                    assert few.solve_kernel() == "winv_gemv_kernel", few.solve_kernel()
                    np.testing.assert_allclose(mu_f, mu_o[:m_few], rtol=MU_RTOL, atol=MU_ATOL)
                    np.testing.assert_allclose(var_f, var_o[:m_few], rtol=0, atol=VAR_ATOL_REL_AMP * amp)
                    ctx.set_tuning("winv_gemv", 0)
                    mu_g, var_g = g.predict(few)
                    assert few.solve_kernel() in ("winv_gemm_kernel", "winv_row_kernel")
                    ctx.set_tuning("winv_gemv", None)
                    np.testing.assert_allclose(mu_f, mu_g, rtol=0, atol=3e-11 * max(1.0, np.abs(mu_g).max()))
                    np.testing.assert_allclose(var_f, var_g, rtol=0, atol=3e-11 * amp)
                    _, _, am_f, _ = g.acq("ei", 0.0, eta, few)
                    assert am_f == int(np.argmax(O.ei(mu_o[:m_few], var_o[:m_few], eta)))
                    few.close()
                mu5, cov5 = g.predict_cov(Xc[:5])
                _, cov5_o = ogp.predict(Xc[:5], full_cov=True)
                np.testing.assert_allclose(np.clip(cov5, np.finfo(float).eps, np.inf), cov5_o, rtol=0,
                                           atol=VAR_ATOL_REL_AMP * amp)
                ctx.set_tuning("winv_rows", rows_mode)
            # chunked passes: same bits
            n_pad = (N + 1 + 127) // 128 * 128
            ctx.set_tuning("ws_bytes", 2 * 128 * n_pad * 8)
            c2 = _lib.Candidates(ctx, Xc)
            mu_c, var_c = g.predict(c2)
            assert c2.solve_kernel() == winv_name and (M <= 256 or c2.chunk() == 256)
            c2.close()
            ctx.set_tuning("ws_bytes", None)
            np.testing.assert_array_equal(mu_c, mu_w)
            np.testing.assert_array_equal(var_c, var_w)
            # consumers of V itself
            mu33, cov = g.predict_cov(Xc[:33])
            _, cov_o = ogp.predict(Xc[:33], full_cov=True)
            np.testing.assert_allclose(np.clip(cov, np.finfo(float).eps, np.inf), cov_o, rtol=0,
                                       atol=VAR_ATOL_REL_AMP * amp)      # (the class clips like the reference; the ABI does not)
            rep = _lib.Candidates(ctx, Xc[40:52] if M > 52 else Xc[:12])
            S = _lib.cross_cov(g, cand, rep)
            from oracle import ig_oracle as IG
            _, S_o = IG.innovation_inputs(ogp, Xc, Xc[40:52] if M > 52 else Xc[:12])
            np.testing.assert_allclose(S, S_o, rtol=0, atol=1e-9 * amp)
            rep.close()
            # a refit at another theta rebuilds W -- here through the asynchronous prefetch (robo_gp_prefetch_inverse:
            # what GaussianProcess.train() issues after its final fit); a second prefetch and a prefetch followed by
            # another fit are harmless
            theta2 = theta.copy()
            theta2[1] += 0.4
            o2 = O.OracleGP(kind, theta2, normalize_input=False)
            o2.train(X, y)
            g.fit(theta, ogp.mean)
            g.prefetch_inverse()                       # (this handle has used W above: the prefetch is live)
            g.fit(theta2, o2.mean)
            g.prefetch_inverse()
            g.prefetch_inverse()
            mu2, var2 = g.predict(cand)
            assert cand.solve_kernel() == winv_name
            g.fit(theta2, o2.mean)
            mu2b, var2b = g.predict(cand)              # built on demand: the same bits
            np.testing.assert_array_equal(mu2, mu2b)
            np.testing.assert_array_equal(var2, var2b)
            m2o, v2o = o2.predict(Xc, diag_only=True)
            np.testing.assert_allclose(mu2, m2o, rtol=MU_RTOL, atol=MU_ATOL)
            np.testing.assert_allclose(var2, v2o, rtol=0, atol=VAR_ATOL_REL_AMP * amp)
            cand.close()
            g.close()
        # conditioning guard: noise 1e-11 on a dense 1-d design -> cond_inf(L) beyond the bound -> substitution
        X = rs.rand(300, 1)
        y = np.sin(4 * X.sum(axis=1))
        theta = np.array([0.0, np.log(0.3 ** 2), np.log(1e-11)])
        g = _lib.DeviceGP(ctx, "matern52", 300, 1)
        g.set_data(X, y)
        g.fit(theta, float(y.mean()))
        cand = _lib.Candidates(ctx, rs.rand(100, 1))
        g.predict(cand)
        assert not cand.solve_kernel().startswith("winv_"), (cand.solve_kernel(), g.factor_cond())
        cand.close()
        g.close()
    finally:
        for key in ("winv_min_blocks", "winv_max", "ws_bytes", "winv_rows", "winv_kc_shift", "winv_gemv"):
            ctx.set_tuning(key, None)


def check_winv_guard_sweep(ctx, n=768, min_blocks=None, m=400, verbose=True,
                           sweep=((2, (1e-3, 1e-5, 1e-7, 1e-9)), (1, (1e-7, 1e-9, 1e-10, 1e-11, 1e-12)))):
    """The guard of the explicit-inverse posterior (api.hip decide_winv): cond_inf(L) = |L|_inf |W|_inf, exact, measured when
    W = L^-1 is built.  Sweep noise x {uniform, clustered-near-incumbent} designs (2-d: the review's noise range; dense
    1-d designs reach the bound):
      * robo_gp_factor_cond equals NumPy's |L|_inf |L^-1|_inf of the factor read back;
      * whichever path the guard picks meets tests/_tol.py against the oracle (mean 1e-10 + 1e-9 |mu|, variance
        1e-8 k(x,x));
      * the substitution is chosen BEFORE the explicit inverse would miss that tolerance: wherever the forced explicit
        inverse misses it, the guard had picked the substitution;
      * beyond the bound the chosen path is the substitution and its result is the forced substitution's, bit for bit.
    Returns the table (design, noise, cond, chosen kernel, errors of the forced paths)."""
    rs = np.random.RandomState(20)
    if min_blocks is not None:
        ctx.set_tuning("winv_min_blocks", min_blocks)
    table = []
    try:
        for D, noises in sweep:
            inc = rs.rand(D)
            designs = {"uniform": rs.rand(n, D),
                       # a quarter space-filling, the rest a cloud around the incumbent (what a BO run's data look like
                       # after RandomSampling's N(incumbent, 0.1) candidates converged: random_sampling.py:43-47)
                       "clustered": np.vstack([rs.rand(n // 4, D), np.clip(inc + 0.02 * rs.randn(n - n // 4, D), 0, 1)])}
            for name, X in designs.items():
                y = np.sin(3 * X.sum(axis=1))
                c = float(y.mean())
                Xs = np.vstack([rs.rand(m // 2, D), np.clip(X[-(m // 2):] + 0.01 * rs.randn(m // 2, D), 0, 1)])
                for noise in noises:
                    theta = np.concatenate([[0.0], np.full(D, np.log(0.25 * D)), [np.log(noise)]])
                    g = _lib.DeviceGP(ctx, "matern52", n, D)
                    g.set_data(X, y)
                    g.fit(theta, c)
                    L = O.gp_compute("matern52", theta, X)
                    mu_o, var_o = O.gp_predict_diag("matern52", theta, L, X, y, c, Xs)
                    cond, dmin, dmax = g.factor_cond()
                    Ld = g.factor()
                    cond_np = np.abs(Ld).sum(axis=1).max() * np.abs(np.linalg.inv(Ld)).sum(axis=1).max()
                    np.testing.assert_allclose(cond, cond_np, rtol=1e-6)
                    assert dmax / dmin <= cond                       # the diagonal ratio only bounds it from below
                    tol_mu = MU_ATOL + MU_RTOL * np.abs(mu_o)
                    tol_var = VAR_ATOL_REL_AMP * np.exp(theta[0])
                    res = {}
                    for path, (key, val) in (("chosen", (None, None)), ("inverse", ("winv_cond_max", 9 * 10 ** 18)),
                                             ("substitution", ("winv_max", 0))):
                        if key:
                            ctx.set_tuning(key, val)
                        try:
                            cand = _lib.Candidates(ctx, Xs)
                            mu, var = g.predict(cand)
                            res[path] = (cand.solve_kernel(), mu, var, bool(np.all(np.abs(mu - mu_o) <= tol_mu) and
                                                                             np.all(np.abs(var - var_o) <= tol_var)))
                            cand.close()
                        finally:
                            if key:
                                ctx.set_tuning(key, None)
                    assert res["inverse"][0].startswith("winv_") and not res["substitution"][0].startswith("winv_")
                    kern, mu, var, ok = res["chosen"]
                    row = (name, D, noise, cond, kern, float(np.abs(res["inverse"][1] - mu_o).max()),
                           float(np.abs(res["substitution"][1] - mu_o).max()),
                           float(np.abs(res["inverse"][2] - var_o).max()),
                           float(np.abs(res["substitution"][2] - var_o).max()))
                    table.append(row)
                    if verbose:
                        print("guard sweep %-9s D=%d noise %.0e cond_inf %.3g -> %-22s |dmu| inverse %.1e substitution %.1e"
                              "  |dvar| %.1e / %.1e" % row)
                    if kern.startswith("winv_"):
                        assert cond <= 1.0e5 and ok, row             # inside the bound AND inside the tolerance
                        np.testing.assert_array_equal(mu, res["inverse"][1])
                    else:
                        assert cond > 1.0e5, row
                        np.testing.assert_array_equal(mu, res["substitution"][1])
                        np.testing.assert_array_equal(var, res["substitution"][2])
                        # beyond the bound the problem's own conditioning is what is left (the factorisation's error,
                        # shared by every path): the substitution stays within 10x the stated tolerance at cond 1e7
                        assert np.all(np.abs(mu - mu_o) <= 10 * tol_mu) and np.all(np.abs(var - var_o) <= tol_var), row
                    if not res["inverse"][3]:
                        # the explicit inverse would have missed the stated tolerance here: the guard must not pick it
                        assert not kern.startswith("winv_"), row
                    g.close()
    finally:
        if min_blocks is not None:
            ctx.set_tuning("winv_min_blocks", None)
    # the sweep must actually exercise both sides of the bound
    assert any(r[4].startswith("winv_") for r in table) and any(not r[4].startswith("winv_") for r in table)
    return table




def check_panel_followers(ctx, sizes=((520, 3), (512, 3), (300, 3), (130, 2)), caps=(64, 16, 13), froms=(-1, 0, 2), emulated=True):
    """potrf_follow: the panel solve of column k+1 inside step k's launch, FOLLOWING the diagonal workgroup through progress
    words (potrf_step_follow_kernel) -- the same factor, likelihood and posterior, BIT FOR BIT, as the launch-per-phase
    form, whichever step the hand-off starts at, however many workgroups share the other tiles, N a multiple of 128 or not.
    (On the CPU the interpreter runs workgroup 0 first, so this checks arithmetic and indexing; the hand-off itself --
    write-through stores, polls, L1-bypassing loads across XCDs -- is what the MI355X run of the same check is for.)"""
    from oracle import gp_oracle as O
    keys = ("potrf_tm4_min", "potrf_max_wg", "potrf_tail_split", "potrf_follow", "potrf_follow_from", "potrf_follow_rows")
    for N, D in sizes:
        rs = np.random.RandomState(13 + N)
        X = rs.rand(N, D)
        y = np.cos(3 * X.sum(axis=1))
        theta = np.concatenate([[0.1], np.log(0.5 + 0.2 * np.arange(D)), [np.log(1e-2)]])
        ogp = O.OracleGP("matern52", theta, normalize_input=False)
        ogp.train(X, y)
        g = _lib.DeviceGP(ctx, "matern52", N, D)
        g.set_data(X, y)
        Xc = rs.rand(40, D)
        try:
            ctx.set_tuning("potrf_follow", 0)
            ll0 = g.fit(theta, ogp.mean)
            L0 = g.factor().copy()
            mu0, v0 = g.predict(Xc)
            np.testing.assert_allclose(ll0, ogp.loglikelihood(theta), rtol=1e-10)
            np.testing.assert_allclose(L0, ogp.L, rtol=0, atol=1e-10)
            if emulated:
                ctx.set_tuning("potrf_tm4_min", 1)
            for frm in froms:
              for frows in (64, 128, -1):              # followers per block row: two, one, by step
                for cap in caps:
                    for split in ((1, 0) if emulated else (1,)):
                        ctx.set_tuning("potrf_follow_rows", frows)
                        ctx.set_tuning("potrf_follow", 1)
                        ctx.set_tuning("potrf_follow_from", frm)
                        if cap is not None:
                            ctx.set_tuning("potrf_max_wg", cap)
                        ctx.set_tuning("potrf_tail_split", split)
                        for rep in range(1 if emulated else 3):
                            ll = g.fit(theta, ogp.mean)
                            assert ll == ll0, (N, frm, frows, cap, split, rep, ll, ll0)
                            np.testing.assert_array_equal(g.factor(), L0, err_msg=str((N, frm, frows, cap, split, rep)))
                        mu, v = g.predict(Xc)
                        np.testing.assert_array_equal(mu, mu0)
                        np.testing.assert_array_equal(v, v0)
        finally:
            for key in keys:
                ctx.set_tuning(key, None)
            g.close()


def check_batched_followers(ctx, sizes=((520, 3, 5), (512, 3, 4), (300, 2, 7)), emulated=True, groups=(0, 1, 2)):
    """potrf_batch_follow: diagonal block + panel of a batched step in ONE launch (potrf_diag_follow_kernel: the x grid index
    is the sample, so all diagonal workgroups are dispatched before any follower; 128-row followers, two strips per wave) --
    likelihoods, kept factors and posteriors of the kept factors equal the launch-per-phase form's BIT FOR BIT, with one,
    two and three sub-batch streams and every group size."""
    keys = ("potrf_batch_follow", "potrf_batch_roll", "potrf_split", "potrf_split_min", "potrf_group", "potrf_batch_tm4_min",
            "potrf_fused_panels", "potrf_max_wg")
    for N, D, S in sizes:
        rs = np.random.RandomState(N)
        X = rs.rand(N, D)
        y = np.cos(3 * X.sum(axis=1))
        th = np.concatenate([[0.1], np.log(0.5 + 0.2 * np.arange(D)), [np.log(1e-2)]])
        thetas = th[None, :] + 0.1 * rs.randn(S, th.size)
        g = _lib.DeviceGP(ctx, "matern52", N, D)
        g.set_data(X, y)
        gps = [_lib.DeviceGP(ctx, "matern52", N, D) for _ in range(S)]
        gps[0].set_data(X, y)
        try:
            ref = None
            for bf in (0, 1, 2):          # launch-per-phase; merged launch; merged launch, 80-KB rolling layout
                for split, smin in ((1, 12), (2, 2), (3, 2)):
                    for grp in (groups if emulated else (0,)):
                        ctx.set_tuning("potrf_fused_panels", 0)
                        ctx.set_tuning("potrf_batch_follow", min(bf, 1))
                        ctx.set_tuning("potrf_batch_roll", 1 if bf == 2 else 0)
                        ctx.set_tuning("potrf_split", split)
                        ctx.set_tuning("potrf_split_min", smin)
                        ctx.set_tuning("potrf_group", grp)
                        if emulated:
                            ctx.set_tuning("potrf_batch_tm4_min", 1)
                        g.loglik_batch(thetas + 0.01, 0.0)         # other matrices through the workspace first
                        for rep in range(1 if emulated else 3):
                            ll, st = g.loglik_batch(thetas, 0.0)
                            assert np.all(st == _lib.OK), st
                            if ref is None:
                                ref = ll.copy()
                            np.testing.assert_array_equal(ll, ref, err_msg=str((N, bf, split, grp, rep)))
            for key in keys:
                ctx.set_tuning(key, None)
            # batches of SMALL factors through the fused step kernels (potrf_fused_panels; default: by residency): same bits
            for fp in (0, 64):
                ctx.set_tuning("potrf_fused_panels", fp)
                if emulated:
                    ctx.set_tuning("potrf_max_wg", 256)
                g.loglik_batch(thetas + 0.01, 0.0)
                ll, st = g.loglik_batch(thetas, 0.0)
                assert np.all(st == _lib.OK), st
                np.testing.assert_array_equal(ll, ref, err_msg="fused batch %d" % fp)
            ctx.set_tuning("potrf_fused_panels", 0)
            ctx.set_tuning("potrf_batch_follow", 0)
            _lib.fit_batch(gps, thetas, 0.0)
            L0 = [h.factor().copy() for h in gps]
            v0 = [h.predict(X[:9] + 0.01) for h in gps]
            for bf, fp in ((1, 0), (0, 64)):          # merged diagonal-block + panel launch; fused step kernels
                ctx.set_tuning("potrf_batch_follow", bf)
                ctx.set_tuning("potrf_fused_panels", fp)
                _lib.fit_batch(gps, thetas, 0.0)
                for h, L, pv in zip(gps, L0, v0):
                    np.testing.assert_array_equal(h.factor(), L)
                    mu, var = h.predict(X[:9] + 0.01)
                    np.testing.assert_array_equal(mu, pv[0])
                    np.testing.assert_array_equal(var, pv[1])
        finally:
            for key in keys:
                ctx.set_tuning(key, None)
            for h in gps:
                h.close()
            g.close()

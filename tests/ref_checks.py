"""Parity of robo_amd's RoBO-surface classes against fixtures produced by the REFERENCE'S OWN classes
(tests/golden/make_golden_ref.py: robo.models.* / robo.acquisition_functions.* / robo.fmin.* executed
unchanged on the george/emcee stand-ins of oracle/refstub).  Every check goes through the product's
public classes and therefore through the C ABI.

Like tests/parity_checks.py the same functions run in two settings: tests/test_ref_parity.py -m gpu
(librobo_hip.so on the MI355X: the parity claim) and, for the small cases, through tests/hipemu on the
CPU (host-logic check only).  Tolerances are tests/_tol.py's stated fp64 tolerances.
"""
import os

import numpy as np

from _tol import ACQ_RTOL, LOGLIK_RTOL, MU_ATOL, MU_RTOL, VAR_ATOL_REL_AMP
import make_golden_ref as G
from robo_amd import acquisition_functions as A
from robo_amd import _lib
from robo_amd.kernels import ExpSquaredKernel, FabolasKernel, Matern52Kernel
from robo_amd.models import FabolasGP, FabolasGPMCMC, GaussianProcess, GaussianProcessMCMC
from robo_amd.priors import DefaultPrior

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
EPS = float(np.finfo(np.float64).eps)


def load(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def _kernel(kind, D, theta_k):
    k = 1.0 * {"matern52": Matern52Kernel, "rbf": ExpSquaredKernel}[kind](np.ones(D), ndim=D)      # (with the amplitude factor)
    k.set_parameter_vector(theta_k)
    return k


def _fabolas_kernel(D, theta_k=None):
    k = FabolasKernel(D + 1)
    if theta_k is not None:
        k.set_parameter_vector(theta_k)
    return k


def _close_acq(vals, ref, mu, var, eta, scale):
    """acquisition values vs the reference classes' outputs where the function is well conditioned in
    (mu, var) (|z| < 8): tolerance of the posterior carried through"""
    z = (eta - mu) / np.sqrt(var)
    well = np.abs(z) < 8
    np.testing.assert_allclose(vals[well], ref[well], rtol=ACQ_RTOL, atol=1e-9 * scale)


def _argmax_ok(am, ref, rtol=ACQ_RTOL):
    """argmax index identical, unless the reference's own top-2 gap is below the value tolerance"""
    want = int(np.argmax(ref))
    srt = np.sort(ref)
    gap = srt[-1] - srt[-2]
    assert am == want or gap <= rtol * abs(ref[want]), (am, want, gap)


# ------------------------------------------------------------------------------------------------
# (1) GaussianProcess  <->  robo/models/gaussian_process.py:70-124,129-191,221-296,334-352
# ------------------------------------------------------------------------------------------------
def check_ref_gp(name, device=None):
    inp, gold = G.ref_inputs(name), load(name)
    D = inp["X"].shape[1]
    gp = GaussianProcess(_kernel(inp["kind"], D, inp["theta"][:-1]), noise=np.exp(inp["theta"][-1]),
                         normalize_output=inp["nout"], lower=inp["lower"], upper=inp["upper"],
                         rng=np.random.RandomState(1), device=device)
    gp.train(inp["X"], inp["y"], do_optimize=False)
    np.testing.assert_allclose(gp.hypers, gold["hypers"], rtol=1e-14)
    assert gp.noise == float(gold["noise"])
    ystd = gp.y_std if inp["nout"] else 1.0
    amp = np.exp(inp["theta"][0]) * ystd ** 2
    scale = max(1.0, np.abs(gold["mu"]).max())
    mu, var = gp.predict(inp["Xc"])
    np.testing.assert_allclose(mu, gold["mu"], rtol=MU_RTOL, atol=MU_ATOL * scale)
    np.testing.assert_allclose(var, gold["var"], rtol=0, atol=VAR_ATOL_REL_AMP * amp)
    inc, inc_val = gp.get_incumbent()
    np.testing.assert_allclose(inc, gold["inc"], rtol=1e-14)
    np.testing.assert_allclose(inc_val, gold["inc_val"], rtol=1e-13)
    _, cov = gp.predict(inp["Xc"][:33], full_cov=True)
    np.testing.assert_allclose(cov, gold["cov33"], rtol=0, atol=VAR_ATOL_REL_AMP * amp)
    pv = gp.predict_variance(inp["Xc"][:1], inp["Xc"][1:20])
    assert pv.shape == gold["pv"].shape
    np.testing.assert_allclose(pv, gold["pv"], rtol=0, atol=VAR_ATOL_REL_AMP * amp)
    eta = float(gold["inc_val"])
    for nm, cls in (("ei", A.EI), ("pi", A.PI), ("lcb", A.LCB)):
        acq = cls(gp)
        vals = acq.compute(inp["Xc"])
        _close_acq(vals, gold[nm], gold["mu"], gold["var"], eta, scale)
        _argmax_ok(acq.argmax(inp["Xc"]), gold[nm])
        assert int(np.argmax(vals)) == acq.argmax(inp["Xc"])
    acq = A.LogEI(gp)
    vals = acq.compute(inp["Xc"])
    core = np.abs((eta - gold["mu"]) / np.sqrt(gold["var"])) < 8
    np.testing.assert_allclose(vals[core], gold["log_ei"][core], rtol=1e-6, atol=1e-9)
    _argmax_ok(acq.argmax(inp["Xc"]), gold["log_ei"], rtol=1e-6)
    if "nll" in gold.files:
        nll = np.array([gp.nll(t) for t in gold["nll_thetas"]])
        np.testing.assert_allclose(nll, gold["nll"], rtol=LOGLIK_RTOL)
        assert nll[3] == 1e25
        for t, ref in zip(gold["nll_thetas"][:3], gold["grad_nll"]):
            np.testing.assert_allclose(gp.grad_nll(t), ref, rtol=1e-7, atol=1e-8 * np.abs(ref).max())
    return gp


def check_ref_gp_retry(device=None):
    """gaussian_process.py:118-122: noise *= 10 exactly once; the second failure escapes train()"""
    gold = load("ref_gp_retry")
    assert str(gold["status"]) == "LinAlgError"
    X = np.random.RandomState(35).rand(200, 2)
    y = G.objective(X)
    theta = gold["theta"]
    gp = GaussianProcess(_kernel("rbf", 2, theta[:-1]), noise=np.exp(theta[-1]), lower=np.zeros(2), upper=np.ones(2),
                         rng=np.random.RandomState(1), device=device)
    raised = False
    try:
        gp.train(X, y, do_optimize=False)
    except np.linalg.LinAlgError:
        raised = True
    assert raised
    np.testing.assert_allclose(gp.noise, float(gold["noise"]), rtol=1e-14)
    assert gp.is_trained == bool(gold["is_trained"])


def check_ref_gp_optimize(device=None):
    """train(do_optimize=True) without a prior (L-BFGS-B on nll with finite differences, :193-219): the
    objective at the reference's optimum is reproduced; the product's own optimum is as good."""
    inp, gold = G.ref_inputs("ref_gp_matern"), load("ref_gp_optimize")
    gp = GaussianProcess(_kernel("matern52", 4, inp["theta"][:-1]), noise=np.exp(inp["theta"][-1]),
                         lower=inp["lower"], upper=inp["upper"], rng=np.random.RandomState(1), device=device)
    gp.train(inp["X"], inp["y"], do_optimize=True)
    mine = gp.nll(gp.hypers)
    np.testing.assert_allclose(gp.nll(gold["hypers"]), float(gold["nll_opt"]), rtol=1e-9)
    np.testing.assert_allclose(gp.nll(inp["theta"]), float(gold["nll_start"]), rtol=LOGLIK_RTOL)
    gain = float(gold["nll_start"]) - float(gold["nll_opt"])
    # finite-difference L-BFGS-B amplifies 1e-13 differences of nll into different iterates: same basin,
    # same objective value to 1e-4 of the improvement the reference achieved
    assert abs(mine - float(gold["nll_opt"])) <= 1e-4 * gain, (mine, float(gold["nll_opt"]))
    return gp.hypers, gold["hypers"]


# ------------------------------------------------------------------------------------------------
# (2) GaussianProcessMCMC + MarginalizationGPMCMC  <->  gaussian_process_mcmc.py:76-166,205-249
# ------------------------------------------------------------------------------------------------
def check_ref_mcmc(device=None):
    inp, gold = G.mcmc_ref_inputs(), load("ref_gpmcmc")
    D = inp["X"].shape[1]
    kernel = 2 * Matern52Kernel(np.ones(D), ndim=D)
    prior = DefaultPrior(len(kernel) + 1, rng=np.random.RandomState(inp["seed"] + 1))
    m = GaussianProcessMCMC(kernel, prior=prior, n_hypers=inp["n_hypers"], chain_length=inp["chain_length"],
                            burnin_steps=inp["burnin_steps"], lower=inp["lower"], upper=inp["upper"],
                            rng=np.random.RandomState(inp["seed"]), device=device)
    m.train(inp["X"], inp["y"], do_optimize=True)
    # the whole chain: prior draws, emcee draw order, batched device likelihoods, accept decisions
    np.testing.assert_allclose(np.array(m.hypers), gold["hypers"], rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(m.p0, gold["p0"], rtol=1e-8, atol=1e-10)
    ll = np.array([m.loglikelihood(h) for h in gold["hypers"]])
    np.testing.assert_allclose(ll, gold["loglik"], rtol=1e-9)
    for s, mm in enumerate(m.models):
        mu, var = mm.predict(inp["Xc"])
        np.testing.assert_allclose(mu, gold["mu_s"][s], rtol=1e-7, atol=1e-8)
        np.testing.assert_allclose(var, gold["var_s"][s], rtol=0, atol=1e-7 * np.exp(gold["hypers"][s][0]))
    mm_, mv_ = m.predict(inp["Xc"])
    np.testing.assert_allclose(mm_, gold["mix_m"], rtol=1e-7, atol=1e-8)
    np.testing.assert_allclose(mv_, gold["mix_v"], rtol=1e-6, atol=1e-9)
    inc, inc_val = m.get_incumbent()
    np.testing.assert_allclose(inc, gold["inc"], rtol=1e-14)
    np.testing.assert_allclose(inc_val, gold["inc_val"], rtol=1e-14)
    for nm, cls, rtol in (("ei", A.EI, 1e-5), ("pi", A.PI, 1e-5), ("lcb", A.LCB, 1e-6), ("log_ei", A.LogEI, 1e-5)):
        acq = A.MarginalizationGPMCMC(cls(m))
        vals = acq.compute(inp["Xc"])
        np.testing.assert_allclose(vals, gold["marg_" + nm], rtol=rtol, atol=1e-9)
        _argmax_ok(acq.argmax(inp["Xc"]), gold["marg_" + nm], rtol=rtol)
    # second BO iteration: burned, walkers continue from p0, both RNG streams continue
    m.train(np.concatenate((inp["X"], inp["X2"])), np.concatenate((inp["y"], inp["y2"])), do_optimize=True)
    np.testing.assert_allclose(np.array(m.hypers), gold["hypers2"], rtol=1e-8, atol=1e-10)
    mm_, mv_ = m.predict(inp["Xc"])
    np.testing.assert_allclose(mm_, gold["mix_m2"], rtol=1e-7, atol=1e-8)
    np.testing.assert_allclose(mv_, gold["mix_v2"], rtol=1e-6, atol=1e-9)
    # do_optimize=False: one model at the kernel's vector with the raw log-noise -8 (:144-147)
    m2 = GaussianProcessMCMC(2 * Matern52Kernel(np.ones(D), ndim=D), lower=inp["lower"], upper=inp["upper"],
                             rng=np.random.RandomState(3), device=device)
    m2.train(inp["X"], inp["y"], do_optimize=False)
    np.testing.assert_allclose(np.array(m2.hypers), gold["noopt_hypers"], rtol=1e-14)
    a, b = m2.predict(inp["Xc"])
    np.testing.assert_allclose(a, gold["noopt_m"], rtol=1e-7, atol=1e-8)
    np.testing.assert_allclose(b, gold["noopt_v"], rtol=0, atol=1e-7)


# ------------------------------------------------------------------------------------------------
# (3) FabolasGP / FabolasGPMCMC  <->  robo/models/fabolas_gp.py
# ------------------------------------------------------------------------------------------------
def _fabolas_mcmc(inp, which, device=None):
    D = inp["D"]
    basis = (lambda x: (1 - x) ** 2) if which == "obj" else (lambda x: x)
    target = inp["y"] if which == "obj" else inp["cost"]
    thetas = inp["thetas"] if which == "obj" else inp["thetas_cost"]
    mc = FabolasGPMCMC(_fabolas_kernel(D), basis_func=basis, n_hypers=len(thetas), lower=inp["lower"],
                       upper=inp["upper"], rng=np.random.RandomState(5), device=device)
    mc.hypers = [t for t in thetas]
    mc.train(inp["X"], target, do_optimize=False)
    return mc


def check_ref_fabolas(device=None):
    inp, gold = G.fabolas_inputs(), load("ref_fabolas")
    D = inp["D"]
    th = inp["thetas"][0]
    gp = FabolasGP(_fabolas_kernel(D, th[:-1]), basis_function=lambda x: (1 - x) ** 2, noise=np.exp(th[-1]),
                   lower=inp["lower"], upper=inp["upper"], rng=np.random.RandomState(1), device=device)
    gp.train(inp["X"], inp["y"], do_optimize=False)
    amp = np.exp(th[0]) * (np.exp(th[-3]) + np.exp(th[-2]))
    mu, var = gp.predict(inp["Xc"])
    np.testing.assert_allclose(mu, gold["mu"], rtol=MU_RTOL, atol=MU_ATOL)
    np.testing.assert_allclose(var, gold["var"], rtol=0, atol=VAR_ATOL_REL_AMP * amp)
    inc, inc_val = gp.get_incumbent()            # projected, double-normalised like the reference
    np.testing.assert_allclose(inc, gold["inc"], rtol=1e-14)
    np.testing.assert_allclose(inc_val, gold["inc_val"], rtol=MU_RTOL, atol=MU_ATOL)
    _, cov = gp.predict(inp["Xc"][:17], full_cov=True)
    np.testing.assert_allclose(cov, gold["cov17"], rtol=0, atol=VAR_ATOL_REL_AMP * amp)
    ei = A.EI(gp).compute(inp["Xc"])
    _close_acq(ei, gold["ei"], gold["mu"], gold["var"], float(gold["inc_val"]), 1.0)
    np.testing.assert_allclose(gp.nll(inp["thetas"][1]), float(gold["nll"]), rtol=LOGLIK_RTOL)
    mc = _fabolas_mcmc(inp, "obj", device)
    mm_, mv_ = mc.predict(inp["Xc"])
    np.testing.assert_allclose(mm_, gold["mix_m"], rtol=1e-8, atol=1e-9)
    np.testing.assert_allclose(mv_, gold["mix_v"], rtol=1e-7, atol=1e-10)
    inc, inc_val = mc.get_incumbent()
    np.testing.assert_allclose(inc, gold["mcmc_inc"], rtol=1e-14)
    np.testing.assert_allclose(inc_val, gold["mcmc_inc_val"], rtol=1e-14)
    # marginalised closed-form acquisitions over Fabolas sub-models (the fused native path must map the
    # candidates through FabolasGP.normalize -- ADVICE r1)
    for nm, cls, rtol in (("ei", A.EI, 1e-6), ("lcb", A.LCB, 1e-7), ("log_ei", A.LogEI, 1e-5)):
        acq = A.MarginalizationGPMCMC(cls(mc))
        np.testing.assert_allclose(acq.compute(inp["Xc"]), gold["marg_" + nm], rtol=rtol, atol=1e-10)
        _argmax_ok(acq.argmax(inp["Xc"]), gold["marg_" + nm], rtol=rtol)


# ------------------------------------------------------------------------------------------------
# (4) InformationGain / InformationGainPerUnitCost  <->  information_gain.py:87-125,153-272,
#     information_gain_per_unit_cost.py:67-106
# ------------------------------------------------------------------------------------------------
IG_RTOL = 1e-6


def _pinned(cls):
    """the class with its representer points taken from the fixture: the reference samples them with an
    emcee sampler seeded from OS entropy (information_gain.py:139-142 passes no rstate0), so they are inputs"""

    class Pinned(cls):
        def sample_representer_points(self):
            self.sampling_acquisition.update(self.model)
            self.zb, self.lmb = self._zb.copy(), self._lmb.copy()

    return Pinned


def _check_ep(ig, gold, sfx=""):
    np.testing.assert_allclose(ig.sn2, float(gold["sn2" + sfx]), rtol=1e-13)
    np.testing.assert_allclose(ig.logP, gold["logP" + sfx], rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose(ig.dlogPdMu, gold["dlogPdMu" + sfx], rtol=1e-5, atol=1e-7 * np.abs(gold["dlogPdMu" + sfx]).max())


def _ig_close(vals, ref):
    np.testing.assert_allclose(vals, ref, rtol=IG_RTOL, atol=IG_RTOL * np.abs(ref).max())
    _argmax_ok(int(np.argmax(vals)), ref, rtol=IG_RTOL)


def check_ref_infogain(device=None):
    inp, gold = G.infogain_inputs(), load("ref_infogain")
    D = inp["X"].shape[1]
    gp = GaussianProcess(_kernel("matern52", D, inp["theta"][:-1]), noise=np.exp(inp["theta"][-1]),
                         lower=inp["lower"], upper=inp["upper"], rng=np.random.RandomState(1), device=device)
    gp.train(inp["X"], inp["y"], do_optimize=False)
    ig = _pinned(A.InformationGain)(gp, inp["lower"], inp["upper"], Nb=50, Np=400, rng=np.random.RandomState(8))
    ig._zb, ig._lmb = gold["zb"], gold["lmb"]
    ig.update(gp)
    _check_ep(ig, gold)
    vals = ig.compute(inp["Xc"])                  # EVERY candidate, from raw coordinates
    _ig_close(vals, gold["ig"])
    assert ig.argmax(inp["Xc"]) == int(gold["argmax"])


def check_ref_infogain_cost(device=None):
    """InformationGainPerUnitCost on FabolasGPMCMC objective + cost models, marginalised over the
    hyper-parameter samples exactly as robo/fmin/fabolas.py:190-198,245 wires it"""
    gold = load("ref_infogain_cost")
    inp = G.fabolas_inputs(M=120)
    D = inp["D"]
    mc_obj, mc_cost = _fabolas_mcmc(inp, "obj", device), _fabolas_mcmc(inp, "cost", device)
    lower, upper = np.append(inp["lower"], 0), np.append(inp["upper"], 1)
    is_env = np.zeros(D + 1)
    is_env[-1] = 1
    S = int(gold["S"])

    class Pinned(A.InformationGainPerUnitCost):
        def sample_representer_points(self):
            self.sampling_acquisition.update(self.model)
            i = [k for k, mdl in enumerate(mc_obj.models) if mdl is self.model][0]
            self.zb, self.lmb = gold["zb_%d" % i].copy(), gold["lmb_%d" % i].copy()

    igc = Pinned(mc_obj, mc_cost, lower, upper, sampling_acquisition=A.EI, is_env_variable=is_env, n_representer=20)
    marg = A.MarginalizationGPMCMC(igc)
    marg.update(mc_obj, mc_cost, overhead=float(gold["overhead"]))
    assert len(marg.estimators) == S
    for i, e in enumerate(marg.estimators):
        _check_ep(e, gold, "_%d" % i)
        np.testing.assert_allclose(mc_cost.models[i].predict(inp["Xc"])[0], gold["log_cost_%d" % i], rtol=1e-8,
                                   atol=1e-9)
        assert e._native_cost()             # gains, cost posterior, division and argmax: ONE library call
        _ig_close(e.compute(inp["Xc"]), gold["ig_%d" % i])
        assert e.argmax(inp["Xc"]) == int(np.argmax(gold["ig_%d" % i]))
        # a candidate outside the box takes the reference's np.spacing(1) / cost branch (host) -- same values elsewhere
        Xo = inp["Xc"].copy()
        Xo[3, 0] = upper[0] + 1.0
        vo = e.compute(Xo)
        keep = np.arange(Xo.shape[0]) != 3
        _ig_close(vo[keep], gold["ig_%d" % i][keep])
        assert 0.0 < vo[3] < 1e-10
    _ig_close(marg.compute(inp["Xc"]), gold["marg"])


def check_ref_infogain_config4(device=None):
    """BASELINE config 4's shape (Fabolas kernel, D = 10 + 1, Nb = 50, Np = 400) against the reference's own
    per-candidate loop at N = 2048"""
    gold = load("ref_infogain_config4")
    N, D, M = 2048, 10, gold["ig"].shape[0]
    inp = G.fabolas_inputs(N=N, D=D, M=M, S=1, seed=71)
    th = inp["thetas"][0]
    gp = FabolasGP(_fabolas_kernel(D, th[:-1]), basis_function=lambda x: (1 - x) ** 2, noise=np.exp(th[-1]),
                   lower=inp["lower"], upper=inp["upper"], rng=np.random.RandomState(1), device=device)
    gp.train(inp["X"], inp["y"], do_optimize=False)
    lower, upper = np.append(inp["lower"], 0), np.append(inp["upper"], 1)
    ig = _pinned(A.InformationGain)(gp, lower, upper, Nb=50, Np=400, sampling_acquisition=A.EI,
                                    rng=np.random.RandomState(12))
    ig._zb, ig._lmb = gold["zb"], gold["lmb"]
    ig.update(gp)
    _check_ep(ig, gold)
    _ig_close(ig.compute(inp["Xc"]), gold["ig"])


# ------------------------------------------------------------------------------------------------
# (5) BASELINE config 1: the reference's 30-iteration Branin run replayed
# ------------------------------------------------------------------------------------------------
def check_ref_branin_replay(device=None):
    """At every iteration of the reference's own run (robo.fmin.bayesian_optimization, GP + EI +
    RandomSampling): same data so far, the hyper-parameters the reference's optimiser found, the global RNG
    state the reference had before maximising -> robo_amd's GaussianProcess + EI + RandomSampling must pick
    the SAME candidate, bit for bit (x_new is a row of the candidate matrix, so equality of the chosen point
    is equality of the argmax index over the 500 candidates)."""
    from robo_amd.maximizers import RandomSampling
    gold = load("ref_branin")
    lo, hi = np.array([-5.0, 0.0]), np.array([10.0, 15.0])
    X, y = gold["X"], gold["y"]
    np.testing.assert_array_equal(y, np.array([G.branin(x) for x in X]))
    kernel = 2 * Matern52Kernel(np.ones(2), ndim=2)
    gp = GaussianProcess(kernel, lower=lo, upper=hi, rng=np.random.RandomState(0), device=device)
    acq = A.EI(gp)
    rs = RandomSampling(acq, lo, hi, rng=np.random.RandomState(0))
    n_checked = 0
    for it, n in enumerate(gold["n"]):
        h = gold["hypers"][it]
        gp.kernel.set_parameter_vector(h[:-1])
        gp.noise = np.exp(h[-1])
        gp.train(X[:n], y[:n], do_optimize=False)
        np.testing.assert_allclose(gp.noise, gold["noise"][it], rtol=1e-12)     # incl. the *10 retry, if any
        acq.update(gp)
        np.random.set_state(("MT19937", gold["rng_keys"][it], int(gold["rng_pos"][it]),
                             int(gold["rng_has_gauss"][it]), float(gold["rng_cached"][it])))
        x_new = rs.maximize()
        np.testing.assert_array_equal(x_new, X[n], err_msg="iteration %d (n=%d)" % (it, n))
        n_checked += 1
    assert n_checked == 27
    return n_checked


def check_ref_branin_free_run(device=None):
    """the same front end left to itself (own L-BFGS-B runs): identical initial design and first model-based
    choice; ends within the reference's regret on Branin"""
    from robo_amd.fmin import bayesian_optimization
    gold = load("ref_branin")
    seed = int(gold["seed"])
    np.random.seed(seed)
    res = bayesian_optimization(G.branin, np.array([-5.0, 0.0]), np.array([10.0, 15.0]), num_iterations=30, n_init=3,
                                model_type="gp", acquisition_func="ei", maximizer="random",
                                rng=np.random.RandomState(seed))
    Xm = np.array(res["X"])
    np.testing.assert_array_equal(Xm[:3], gold["X"][:3])
    same = 0
    while same < 30 and np.array_equal(Xm[same], gold["X"][same]):
        same += 1
    return same, float(res["f_opt"]), float(gold["f_opt"])


def check_ref_single_point_replay(device=None, which=("scipy", "differential_evolution"), max_iters=None):
    """At every model-based iteration of the reference's own runs of robo.fmin.bayesian_optimization(maximizer="scipy" /
    "differential_evolution") on Branin (fixture ref_branin_single_point, tests/golden/make_golden_ref.py): same data so far,
    the hyper-parameters the reference found, the global RNG state it had before maximising -> robo_amd's
    SciPyOptimizer / DifferentialEvolution over robo_amd's EI.  L-BFGS-B differentiates the acquisition by finite
    differences of 1e-8, so a trajectory cannot be bit-identical across two implementations of EI (a relative 1e-9 in a value is
    10 % of such a difference); what is pinned: the start points are the reference's (same stream consumption: checked
    through the end state of the global stream where the search itself draws nothing), the point found scores at least
    (1 - 1e-3) of what the reference's point scores under the SAME acquisition function, and in most iterations it IS the
    reference's point to 1e-3 of the box.  -> (iterations checked, iterations landing on the reference's point)"""
    from robo_amd.maximizers import DifferentialEvolution, SciPyOptimizer
    gold = load("ref_branin_single_point")
    lo, hi = np.array([-5.0, 0.0]), np.array([10.0, 15.0])
    checked = same = 0
    for name in which:
        X, y = gold[name + "_X"], gold[name + "_y"]
        np.testing.assert_array_equal(y, np.array([G.branin(x) for x in X]))
        kernel = 2 * Matern52Kernel(np.ones(2), ndim=2)
        gp = GaussianProcess(kernel, lower=lo, upper=hi, rng=np.random.RandomState(0), device=device)
        acq = A.EI(gp)
        cls = SciPyOptimizer if name == "scipy" else DifferentialEvolution
        maxi = cls(acq, lo, hi, rng=np.random.RandomState(0))
        for it, n in enumerate(gold[name + "_n"][:max_iters]):
            h = gold[name + "_hypers"][it]
            gp.kernel.set_parameter_vector(h[:-1])
            gp.noise = np.exp(h[-1])
            gp.train(X[:n], y[:n], do_optimize=False)
            np.testing.assert_allclose(gp.noise, gold[name + "_noise"][it], rtol=1e-12)
            acq.update(gp)
            np.random.set_state(("MT19937", gold[name + "_rng_keys"][it], int(gold[name + "_rng_pos"][it]),
                                 int(gold[name + "_rng_has_gauss"][it]), float(gold[name + "_rng_cached"][it])))
            x_new = maxi.maximize()
            x_ref = X[n]
            assert np.all(x_new >= lo) and np.all(x_new <= hi)
            a_new, a_ref = float(acq(x_new[None, :])[0]), float(acq(x_ref[None, :])[0])
            assert a_new >= (1.0 - 1e-3) * a_ref - 1e-12, (name, it, a_new, a_ref)
            same += bool(np.all(np.abs(x_new - x_ref) <= 1e-3 * (hi - lo)))
            if name == "scipy" and it + 1 < len(gold[name + "_n"]):
                # the search draws nothing: the global stream must stand where the reference's stood after ITS maximisation,
                # i.e. before the (deterministic) objective call and the next train -- which draw nothing either
                st = np.random.get_state()
                assert int(st[2]) == int(gold[name + "_rng_pos"][it + 1]) and \
                    np.array_equal(st[1], gold[name + "_rng_keys"][it + 1]), "start points consumed another stream"
            checked += 1
    return checked, same


# ------------------------------------------------------------------------------------------------
# (6) the other two front ends replayed: robo/fmin/entropy_search.py:20-131 (model="gp") and
#     robo/fmin/fabolas.py:31-312, objects wired by robo_amd's own front-end builders
# ------------------------------------------------------------------------------------------------
def _set_global_rng(gold, sfx):
    np.random.set_state(("MT19937", gold["rng_keys" + sfx], int(gold["rng_pos" + sfx]),
                         int(gold["rng_has_gauss" + sfx]), float(gold["rng_cached" + sfx])))


def check_ref_entropy_search_replay(device=None):
    """At every model-based iteration of the reference's own entropy_search run: same data so far, the
    hyper-parameters its L-BFGS-B found, the representer points its emcee sampler drew (seeded from OS entropy
    there: inputs), the global RNG state it had before RandomSampling.maximize -> robo_amd's GaussianProcess +
    InformationGain + RandomSampling (as robo_amd.fmin.entropy_search wires them) must choose the SAME candidate,
    bit for bit, i.e. the same argmax over the 500 information gains."""
    from robo_amd.fmin.entropy_search import build_entropy_search
    gold = load("ref_entropy_search")
    lo, hi = np.array([-5.0, 0.0]), np.array([10.0, 15.0])
    X, y = gold["X"], gold["y"]
    np.testing.assert_array_equal(y, np.array([G.es_objective(x) for x in X]))
    gp, acq, rs = build_entropy_search(lo, hi, "random", "gp", np.random.RandomState(0))
    if device is not None:
        gp.device = device
    for it, n in enumerate(gold["n"]):
        h = gold["hypers"][it]
        gp.kernel.set_parameter_vector(h[:-1])
        gp.noise = np.exp(h[-1])
        gp.train(X[:n], y[:n], do_optimize=False)
        np.testing.assert_allclose(gp.noise, gold["noise"][it], rtol=1e-12)

        def pinned(zb=gold["zb"][it], lmb=gold["lmb"][it]):
            acq.sampling_acquisition.update(acq.model)
            acq.zb, acq.lmb = zb.copy(), lmb.copy()

        acq.sample_representer_points = pinned
        acq.update(gp)
        np.random.set_state(("MT19937", gold["rng_keys"][it], int(gold["rng_pos"][it]),
                             int(gold["rng_has_gauss"][it]), float(gold["rng_cached"][it])))
        x_new = rs.maximize()
        np.testing.assert_array_equal(x_new, gold["x_new"][it], err_msg="iteration %d (n=%d)" % (it, n))
        np.testing.assert_array_equal(x_new, X[n])
        inc, inc_val = gp.get_incumbent()
        np.testing.assert_allclose(inc, gold["incumbents"][n - 1], rtol=1e-13)      # best of the first n points
        np.testing.assert_allclose(inc_val, gold["incumbent_values"][n - 1], rtol=1e-13)
    return len(gold["n"])


def check_ref_fabolas_replay(device=None, n_iter=None):
    """The same for robo.fmin.fabolas: per model-based iteration both models' hyper-parameter samples, every
    estimator's representer points and the global RNG state are the reference run's; the projected incumbent
    (fabolas.py:225-227) and the candidate RandomSampling picks on the information gain per unit cost,
    marginalised over the 12 samples, must be the reference's."""
    from robo_amd.acquisition_functions import InformationGainPerUnitCost
    from robo_amd.fmin.fabolas import build_fabolas
    from robo_amd.util.incumbent_estimation import projected_incumbent_estimation
    gold = load("ref_fabolas_frontend")
    X, y, c, n0, S = gold["X"], gold["y"], gold["c"], int(gold["n0"]), int(gold["S"])
    m_obj, m_cost, acq, rs = build_fabolas(np.zeros(2), np.ones(2), burnin=20, chain_length=10, n_hypers=12,
                                           rng=np.random.RandomState(0))
    assert m_obj.n_hypers == S
    if device is not None:
        m_obj.device = m_cost.device = device
    state = {}

    def pinned(self):
        self.sampling_acquisition.update(self.model)
        i = [k for k, mdl in enumerate(m_obj.models) if mdl is self.model][0]
        self.zb, self.lmb = state["zb"][i].copy(), state["lmb"][i].copy()

    orig = InformationGainPerUnitCost.sample_representer_points
    InformationGainPerUnitCost.sample_representer_points = pinned
    try:
        n_all = X.shape[0] - n0
        for it in range(n_all if n_iter is None else min(n_iter, n_all)):
            n = n0 + it
            m_obj.hypers = [h for h in gold["hypers_obj_%d" % it]]
            m_cost.hypers = [h for h in gold["hypers_cost_%d" % it]]
            m_obj.train(X[:n], y[:n], do_optimize=False)
            m_cost.train(X[:n], c[:n], do_optimize=False)
            inc, inc_val = projected_incumbent_estimation(m_obj, X[:n, :-1], proj_value=1)
            np.testing.assert_allclose(inc, gold["inc_%d" % it], rtol=1e-13)
            np.testing.assert_allclose(inc_val, float(gold["inc_val_%d" % it]), rtol=1e-7, atol=1e-9)
            state["zb"], state["lmb"] = gold["zb_%d" % it], gold["lmb_%d" % it]
            acq.update(m_obj, m_cost)
            assert len(acq.estimators) == S
            _set_global_rng(gold, "_%d" % it)
            x_new = rs.maximize()
            np.testing.assert_array_equal(x_new, gold["x_new_%d" % it], err_msg="iteration %d" % it)
            np.testing.assert_array_equal(x_new, X[n])
        m_obj.hypers = [h for h in gold["hypers_final"]]
        m_obj.train(X, y, do_optimize=False)
        inc, inc_val = projected_incumbent_estimation(m_obj, X[:, :-1], proj_value=1)
        np.testing.assert_allclose(inc[:-1], gold["x_opt"], rtol=1e-13)
        np.testing.assert_allclose(inc_val, float(gold["inc_val_final"]), rtol=1e-7, atol=1e-9)
    finally:
        InformationGainPerUnitCost.sample_representer_points = orig
    return n_all if n_iter is None else min(n_iter, n_all)


def check_ref_branin_gpmcmc_replay(device=None, devices=None, max_iters=None, chain=True):
    """robo.fmin.bayesian_optimization(model_type="gp_mcmc", acquisition_func="log_ei", maximizer="random") -- the
    reference's own run on Branin with the front end's MCMC configuration (DefaultPrior, 10 walkers, 100 burn-in + 200
    steps per iteration; fixture ref_branin_gpmcmc).  At every model-based iteration: the data so far, the walkers' last
    positions the reference's chain ended on, the global RNG state it had before maximising -> robo_amd's
    GaussianProcessMCMC (one batched fit of the 10 samples) + MarginalizationGPMCMC(LogEI) + RandomSampling must pick the
    SAME candidate, bit for bit (= the same argmax of the marginal LogEI over the 500 candidates).
    ``chain``: additionally robo_amd's OWN chains, started from the reference's generator state before the first training
    (prior draw + burn-in + 200 steps, then 200 steps from the previous positions at every later iteration), must end on
    the reference's walkers at EVERY iteration (rtol 1e-6) with the generator standing exactly where the reference's stood.
    -> (iterations checked, smallest relative gap between the best and the second-best candidate)"""
    from robo_amd.maximizers import RandomSampling
    gold = load("ref_branin_gpmcmc")
    lo, hi = np.array([-5.0, 0.0]), np.array([10.0, 15.0])
    X, y = gold["X"], gold["y"]
    np.testing.assert_array_equal(y, np.array([G.branin(x) for x in X]))
    S = gold["hypers"].shape[1]
    assert S == 10 and gold["hypers"].shape[2] == 4      # 3 * len(kernel) = 9 -> even (bayesian_optimization.py:86-88)

    class Pinned(GaussianProcessMCMC):
        def _keep_hypers_without_optimize(self):          # train(do_optimize=False) keeps the samples put in place
            return True

    def build(cls, rng):
        kernel = 2 * Matern52Kernel(np.ones(2), ndim=2)
        kw = dict(devices=devices) if devices is not None else dict(device=device)
        # the front end gives the prior NO generator (bayesian_optimization.py:84): DefaultPrior seeds its own from the
        # GLOBAL stream (default_priors.py:11-12) -- the first global draw after the caller's np.random.seed
        return cls(kernel, prior=DefaultPrior(len(kernel) + 1), n_hypers=S, chain_length=200, burnin_steps=100,
                   normalize_input=True, normalize_output=False, rng=rng, lower=lo, upper=hi, **kw)

    np.random.seed(int(gold["seed"]))
    model = build(Pinned, np.random.RandomState(0))
    acq = A.MarginalizationGPMCMC(A.LogEI(model))
    rs = RandomSampling(acq, lo, hi, rng=np.random.RandomState(0))
    n_checked, gap = 0, np.inf
    for it, n in enumerate(gold["n"]):
        if max_iters is not None and it >= max_iters:
            break
        model.hypers = [h for h in gold["hypers"][it]]
        model.train(X[:n], y[:n], do_optimize=False)
        assert len(model.models) == S
        acq.update(model)
        _set_global_rng({k: gold[k][it] for k in ("rng_keys", "rng_pos", "rng_has_gauss", "rng_cached")}, "")
        cand = rs.candidates()
        vals = np.asarray(acq.compute(cand)).reshape(-1)
        order = np.argsort(vals)
        gap = min(gap, float((vals[order[-1]] - vals[order[-2]]) / abs(vals[order[-1]])))
        _set_global_rng({k: gold[k][it] for k in ("rng_keys", "rng_pos", "rng_has_gauss", "rng_cached")}, "")
        x_new = rs.maximize()
        np.testing.assert_array_equal(x_new, X[n], err_msg="iteration %d (n=%d)" % (it, n))
        np.testing.assert_array_equal(x_new, cand[order[-1]])
        n_checked += 1
    if chain:
        # the chain itself, first training: the reference's generator stood at own_before; robo_amd draws the prior sample,
        # burns in and runs the chain with the same stream consumption and must end where the reference ended
        rng = np.random.RandomState(0)
        rng.set_state(("MT19937", gold["own_before_keys"][0], int(gold["own_before_pos"][0]), 0, 0.0))
        np.random.seed(int(gold["seed"]))
        own = build(GaussianProcessMCMC, rng)
        for it, n in enumerate(gold["n"]):
            if max_iters is not None and it >= max_iters:
                break
            # (nothing else draws from this generator between two trainings: the maximiser uses the global stream)
            own.train(X[:n], y[:n], do_optimize=True)
            np.testing.assert_allclose(np.asarray(own.hypers), gold["hypers"][it], rtol=1e-6, atol=1e-8,
                                       err_msg="chain of iteration %d" % it)
            st = rng.get_state()
            np.testing.assert_array_equal(st[1], gold["own_keys"][it])
            assert int(st[2]) == int(gold["own_pos"][it])
    return n_checked, gap


def check_ref_branin_gpmcmc_free_run(num_iterations=None, acquisition_func="log_ei", **kw):
    """robo_amd.fmin.bayesian_optimization(model_type="gp_mcmc", acquisition_func="log_ei", maximizer="random") LEFT TO
    ITSELF with the seeds of the reference's run: nothing in this configuration is decided by a finite-difference
    optimiser (the hyper-parameters come from the ensemble sampler, the candidate from an argmax over 500 points), so the
    whole result must be the reference's -- every evaluated point bit for bit, hence y, the incumbent trajectory, x_opt and
    f_opt.  (With model_type="gp" the free run diverges after the first differing L-BFGS-B run: check_ref_branin_free_run.)
    -> number of evaluated points compared"""
    from robo_amd.fmin import bayesian_optimization
    if acquisition_func == "log_ei":
        gold = load("ref_branin_gpmcmc")
    else:
        # fixture ref_branin_gpmcmc_acq: the reference's runs with EI / PI / LCB under MarginalizationGPMCMC (8 iterations each)
        full = load("ref_branin_gpmcmc_acq")
        gold = {k[len(acquisition_func) + 1:]: full[k] for k in full.files if k.startswith(acquisition_func + "_")}
    seed = int(gold["seed"])
    n_all = gold["X"].shape[0]
    n_it = n_all if num_iterations is None else int(num_iterations)
    np.random.seed(seed)
    # log_ei: every choice left to the front end's DEFAULTS (the reference's: gp_mcmc, log_ei, random, n_init 3)
    explicit = {} if acquisition_func == "log_ei" else dict(model_type="gp_mcmc", acquisition_func=acquisition_func,
                                                            maximizer="random", n_init=3)
    res = bayesian_optimization(G.branin, np.array([-5.0, 0.0]), np.array([10.0, 15.0]), num_iterations=n_it,
                                rng=np.random.RandomState(seed), **explicit, **kw)
    Xm = np.array(res["X"])
    assert Xm.shape[0] == n_it
    np.testing.assert_array_equal(Xm, gold["X"][:n_it])
    np.testing.assert_array_equal(np.array(res["y"]), gold["y"][:n_it])
    np.testing.assert_array_equal(np.array(res["incumbent_values"]), gold["incumbent_values"][:n_it])
    if n_it == n_all:
        np.testing.assert_array_equal(np.asarray(res["x_opt"]), gold["x_opt"])
        assert float(res["f_opt"]) == float(gold["f_opt"])
    return n_it


def check_ref_entropy_search_gpmcmc_replay(device=None, devices=None, max_iters=None):
    """robo.fmin.entropy_search with its DEFAULT model ("gp_mcmc"; fixture ref_entropy_search_gpmcmc): at every model-based
    iteration of the reference's own run -- the data so far, the 10 walkers the reference's chain ended on, EVERY
    estimator's representer points (OS-entropy seeded in the reference: inputs), the global RNG state before
    RandomSampling.maximize -> robo_amd's GaussianProcessMCMC + MarginalizationGPMCMC(InformationGain) + RandomSampling, as
    robo_amd.fmin.entropy_search wires them, must choose the SAME candidate, bit for bit (the argmax over 500 candidates of
    the information gain averaged over the 10 samples, each with its own EP and innovations).  -> iterations checked"""
    from robo_amd.acquisition_functions import InformationGain
    from robo_amd.fmin.entropy_search import build_entropy_search
    gold = load("ref_entropy_search_gpmcmc")
    lo, hi = np.array([-5.0, 0.0]), np.array([10.0, 15.0])
    X, y = gold["X"], gold["y"]
    np.testing.assert_array_equal(y, np.array([G.es_objective(x) for x in X]))
    S = gold["hypers"].shape[1]
    np.random.seed(int(gold["seed"]))
    model, acq, rs = build_entropy_search(lo, hi, "random", "gp_mcmc", np.random.RandomState(0), devices=devices)
    assert model.n_hypers == S
    if device is not None:
        model.device = device
    model._keep_hypers_without_optimize = lambda: True       # train(do_optimize=False) keeps the samples put in place
    state = {}

    def pinned(self):
        self.sampling_acquisition.update(self.model)
        i = [k for k, mdl in enumerate(model.models) if mdl is self.model][0]
        self.zb, self.lmb = state["zb"][i].copy(), state["lmb"][i].copy()

    orig = InformationGain.sample_representer_points
    InformationGain.sample_representer_points = pinned
    n_checked = 0
    try:
        for it, n in enumerate(gold["n"]):
            if max_iters is not None and it >= max_iters:
                break
            model.hypers = [h for h in gold["hypers"][it]]
            model.train(X[:n], y[:n], do_optimize=False)
            assert len(model.models) == S
            state["zb"], state["lmb"] = gold["zb"][it], gold["lmb"][it]
            acq.update(model)
            assert len(acq.estimators) == S
            _set_global_rng({k: gold[k][it] for k in ("rng_keys", "rng_pos", "rng_has_gauss", "rng_cached")}, "")
            x_new = rs.maximize()
            np.testing.assert_array_equal(x_new, gold["x_new"][it], err_msg="iteration %d (n=%d)" % (it, n))
            np.testing.assert_array_equal(x_new, X[n])
            n_checked += 1
    finally:
        InformationGain.sample_representer_points = orig
    return n_checked

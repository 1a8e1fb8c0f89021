#!/usr/bin/env python
"""Golden fixtures produced by the REFERENCE'S OWN GP-side classes (run in the BUILD container only).

    python tests/golden/make_golden_ref.py [case ...]

``robo.models.{gaussian_process,gaussian_process_mcmc,fabolas_gp}``,
``robo.acquisition_functions.{information_gain,information_gain_per_unit_cost,marginalization}``,
``robo.maximizers.random_sampling``, ``robo.solver.bayesian_optimization`` and
``robo.fmin.bayesian_optimization`` are imported from /root/reference and executed UNCHANGED; the two
third-party packages they need and this image lacks are served by the test-only stand-ins under
oracle/refstub (``george``: the API slice of SURVEY.md A.1 on the stated kernel contract;
``emcee``: the 2.x ensemble sampler restated with its draw order).  What these fixtures pin is
therefore everything the reference's Python does -- normalisation, mean, noise retry, clipping,
mixtures, the Fabolas basis/projection (incl. its double normalisation), innovations, entropy
change, cost division, the solver loop -- on top of the george kernel formulas, which stay the
project's stated contract (george's source is not in the tree).

Only outputs and seeds are stored; ``ref_inputs`` below regenerates the inputs (imported by the tests).
/root/reference does not exist on the GPU box: tests read the committed .npz files only.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from tests.golden.make_golden import objective, default_theta  # noqa: E402


def branin(x):
    x1, x2 = x
    return (x2 - 5.1 * x1 ** 2 / (4 * np.pi ** 2) + 5 * x1 / np.pi - 6) ** 2 + \
        10 * (1 - 1 / (8 * np.pi)) * np.cos(x1) + 10


# ------------------------------------------------------------------------------------------------
# inputs (shared with the tests)
# ------------------------------------------------------------------------------------------------
GP_CASES = {
    # name: kind, N, D, M, lo, hi, normalize_output, seed
    "ref_gp_matern": dict(kind="matern52", N=90, D=4, M=300, lo=-2.0, hi=3.0, nout=False, seed=31),
    "ref_gp_rbf_nout": dict(kind="rbf", N=140, D=3, M=260, lo=0.0, hi=1.0, nout=True, seed=32),
    "ref_gp_headline_shape": dict(kind="matern52", N=1500, D=16, M=2048, lo=0.0, hi=1.0, nout=False, seed=33),
}


def ref_inputs(case):
    c = GP_CASES[case]
    rng = np.random.RandomState(c["seed"])
    X01 = rng.rand(c["N"], c["D"])
    y = objective(X01)
    y = (y - y.mean()) / y.std() * 1.7 + 0.4          # not standardised: exercises mean / output normalisation
    lower, upper = np.full(c["D"], c["lo"]), np.full(c["D"], c["hi"])
    X = lower + (upper - lower) * X01
    Xc = lower + (upper - lower) * np.random.RandomState(c["seed"] + 1000).rand(c["M"], c["D"])
    theta = default_theta(c["kind"], c["D"])
    theta[1:-1] += 0.25 * np.random.RandomState(c["seed"] + 2000).randn(c["D"])
    theta[0] = 0.3
    return dict(X=X, y=y, Xc=Xc, theta=theta, lower=lower, upper=upper, kind=c["kind"], nout=c["nout"])


def mcmc_ref_inputs():
    rng = np.random.RandomState(41)
    N, D, M = 40, 3, 200
    lower, upper = np.array([-1.0, 0.0, 2.0]), np.array([1.0, 5.0, 3.0])
    X = lower + (upper - lower) * rng.rand(N, D)
    y = objective((X - lower) / (upper - lower))
    Xc = lower + (upper - lower) * np.random.RandomState(42).rand(M, D)
    X2 = lower + (upper - lower) * np.random.RandomState(43).rand(3, D)
    y2 = objective((X2 - lower) / (upper - lower))
    return dict(X=X, y=y, Xc=Xc, X2=X2, y2=y2, lower=lower, upper=upper, n_hypers=10, chain_length=12,
                burnin_steps=14, seed=44)


def fabolas_inputs(N=60, D=3, M=150, S=4, seed=51):
    """config-4 shaped: D configuration columns + the dataset-size column s in (0, 1]"""
    rng = np.random.RandomState(seed)
    lower, upper = np.full(D, -1.0), np.full(D, 2.0)
    Xcfg = lower + (upper - lower) * rng.rand(N, D)
    s = rng.rand(N) * 0.95 + 0.05
    X = np.concatenate((Xcfg, s[:, None]), axis=1)
    y = objective((Xcfg - lower) / (upper - lower)) / D + 0.5 * (1 - s) ** 2 + 0.05 * rng.randn(N)
    cost = np.log(0.2 + 3.0 * s) + 0.02 * rng.randn(N)                       # the cost model sees LOG cost
    rc = np.random.RandomState(seed + 1)
    Xc = np.concatenate((lower + (upper - lower) * rc.rand(M, D), rc.rand(M, 1) * 0.95 + 0.05), axis=1)
    P = 1 + D + 2 + 1
    base = np.concatenate([[np.log(1.0 / (D + 1))], np.full(D, np.log(0.4)), [0.1, 0.1], [np.log(1e-3)]])
    thetas = base[None, :] + 0.25 * np.random.RandomState(seed + 2).randn(S, P)
    thetas_cost = base[None, :] + 0.25 * np.random.RandomState(seed + 3).randn(S, P)
    return dict(X=X, y=y, cost=cost, Xc=Xc, lower=lower, upper=upper, thetas=thetas, thetas_cost=thetas_cost, D=D)


def infogain_inputs(N=80, D=2, M=400, seed=61):
    rng = np.random.RandomState(seed)
    lower, upper = np.array([-5.0, 0.0])[:D], np.array([10.0, 15.0])[:D]
    X = lower + (upper - lower) * rng.rand(N, D)
    y = np.array([branin(x) for x in X]) / 50.0
    Xc = lower + (upper - lower) * np.random.RandomState(seed + 1).rand(M, D)
    theta = np.array([0.5, np.log(0.1), np.log(0.15), np.log(1e-3)])
    return dict(X=X, y=y, Xc=Xc, lower=lower, upper=upper, theta=theta)


# ------------------------------------------------------------------------------------------------
# the reference, importable through the stand-ins
# ------------------------------------------------------------------------------------------------
def reference():
    for attr, val in (("Infinity", np.inf), ("NAN", np.nan)):
        if not hasattr(np, attr):
            setattr(np, attr, val)          # NumPy-2 hazards in log_ei.py / epmgp.py
    stub = os.path.join(ROOT, "oracle", "refstub")
    for p in (stub, "/root/reference"):
        if p not in sys.path:
            sys.path.insert(0, p)
    import types
    import george                                    # noqa: F401  (the stand-in)
    import emcee                                     # noqa: F401  (the stand-in)
    assert "refstub" in george.__file__ and "refstub" in emcee.__file__
    ns = types.SimpleNamespace()
    ns.george = george
    from robo.models.gaussian_process import GaussianProcess
    from robo.models.gaussian_process_mcmc import GaussianProcessMCMC
    from robo.models.fabolas_gp import FabolasGP, FabolasGPMCMC
    from robo.acquisition_functions.ei import EI
    from robo.acquisition_functions.log_ei import LogEI
    from robo.acquisition_functions.pi import PI
    from robo.acquisition_functions.lcb import LCB
    from robo.acquisition_functions.marginalization import MarginalizationGPMCMC
    from robo.acquisition_functions.information_gain import InformationGain
    from robo.acquisition_functions.information_gain_per_unit_cost import InformationGainPerUnitCost
    from robo.priors.default_priors import DefaultPrior
    from robo.priors.env_priors import EnvPrior
    ns.__dict__.update(locals())
    return ns


def george_kernel(R, kind, D, theta_k):
    K = {"matern52": R.george.kernels.Matern52Kernel, "rbf": R.george.kernels.ExpSquaredKernel}[kind]
    k = 1.0 * K(np.ones(D), ndim=D)
    k.set_parameter_vector(theta_k)
    return k


def george_fabolas_kernel(R, D, theta_k=None):
    """robo/fmin/fabolas.py:103-117, verbatim construction"""
    kernel = 1
    for d in range(D):
        kernel *= R.george.kernels.Matern52Kernel(np.ones([1]) * 0.01, ndim=D + 1, axes=d)
    kernel *= R.george.kernels.BayesianLinearRegressionKernel(log_a=0.1, log_b=0.1, ndim=D + 1, axes=D)
    if theta_k is not None:
        kernel.set_parameter_vector(theta_k)
    return kernel


def _save(name, **out):
    np.savez(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name, sorted(out))


# ------------------------------------------------------------------------------------------------
# (1) GaussianProcess.train / predict / nll / grad_nll / predict_variance / get_incumbent
# ------------------------------------------------------------------------------------------------
def make_gp(R):
    for name, c in GP_CASES.items():
        inp = ref_inputs(name)
        D = c["D"]
        kernel = george_kernel(R, c["kind"], D, inp["theta"][:-1])
        gp = R.GaussianProcess(kernel, noise=np.exp(inp["theta"][-1]), normalize_output=c["nout"],
                               lower=inp["lower"], upper=inp["upper"], rng=np.random.RandomState(1))
        gp.train(inp["X"], inp["y"], do_optimize=False)
        mu, var = gp.predict(inp["Xc"])
        out = dict(mu=mu, var=var, hypers=np.array(gp.hypers), noise=gp.noise)
        inc, inc_val = gp.get_incumbent()
        out.update(inc=inc, inc_val=inc_val)
        _, cov = gp.predict(inp["Xc"][:33], full_cov=True)
        out["cov33"] = cov
        out["pv"] = gp.predict_variance(inp["Xc"][:1], inp["Xc"][1:20])
        for nm, cls in (("ei", R.EI), ("log_ei", R.LogEI), ("pi", R.PI), ("lcb", R.LCB)):
            out[nm] = cls(gp).compute(inp["Xc"])
            out["argmax_" + nm] = int(np.argmax(out[nm]))
        if c["N"] <= 256:
            thetas = inp["theta"][None, :] + 0.4 * np.random.RandomState(c["seed"] + 3000).randn(4, D + 2)
            thetas[3, 1] = 25.0                      # out of the |theta| <= 20 box -> 1e25
            out["nll_thetas"] = thetas
            out["nll"] = np.array([gp.nll(t) for t in thetas])
            out["grad_nll"] = np.array([gp.grad_nll(t) for t in thetas[:3]])
        _save(name, **out)

    # noise * 10 retry (gaussian_process.py:118-122).  A numerically rank-deficient K (smooth RBF, long length
    # scales, 200 points, amplitude e^8) with noise 1e-18: gp.compute fails, the retry at 1e-17 fails as well and
    # the LinAlgError of the SECOND compute escapes train(); noise has been multiplied by 10 exactly once.
    rng = np.random.RandomState(35)
    X = rng.rand(200, 2)
    y = objective(X)
    theta = np.array([8.0, np.log(30.0), np.log(30.0), np.log(1e-18)])
    gp = R.GaussianProcess(george_kernel(R, "rbf", 2, theta[:-1]), noise=np.exp(theta[-1]), lower=np.zeros(2),
                           upper=np.ones(2), rng=np.random.RandomState(1))
    status = "ok"
    try:
        gp.train(X, y, do_optimize=False)
    except np.linalg.LinAlgError:
        status = "LinAlgError"
    _save("ref_gp_retry", status=status, noise=gp.noise, is_trained=gp.is_trained, theta=theta)

    # do_optimize=True without a prior (L-BFGS-B on nll, finite differences, gaussian_process.py:193-219)
    inp = ref_inputs("ref_gp_matern")
    D = 4
    kernel = george_kernel(R, "matern52", D, inp["theta"][:-1])
    gp = R.GaussianProcess(kernel, noise=np.exp(inp["theta"][-1]), lower=inp["lower"], upper=inp["upper"],
                           rng=np.random.RandomState(1))
    gp.train(inp["X"], inp["y"], do_optimize=True)
    _save("ref_gp_optimize", hypers=gp.hypers, nll_opt=gp.nll(gp.hypers), nll_start=gp.nll(inp["theta"]))


# ------------------------------------------------------------------------------------------------
# (2) GaussianProcessMCMC.train (emcee draw order) / predict + MarginalizationGPMCMC
# ------------------------------------------------------------------------------------------------
def make_mcmc(R):
    inp = mcmc_ref_inputs()
    D = inp["X"].shape[1]
    kernel = 2 * R.george.kernels.Matern52Kernel(np.ones(D), ndim=D)     # fmin/bayesian_optimization.py:75-81
    prior = R.DefaultPrior(len(kernel) + 1, rng=np.random.RandomState(inp["seed"] + 1))
    m = R.GaussianProcessMCMC(kernel, prior=prior, n_hypers=inp["n_hypers"], chain_length=inp["chain_length"],
                              burnin_steps=inp["burnin_steps"], lower=inp["lower"], upper=inp["upper"],
                              rng=np.random.RandomState(inp["seed"]))
    m.train(inp["X"], inp["y"], do_optimize=True)
    out = dict(hypers=np.array(m.hypers), p0=np.array(m.p0))
    out["mix_m"], out["mix_v"] = m.predict(inp["Xc"])
    out["mu_s"] = np.array([mm.predict(inp["Xc"])[0] for mm in m.models])
    out["var_s"] = np.array([mm.predict(inp["Xc"])[1] for mm in m.models])
    out["loglik"] = np.array([m.loglikelihood(h) for h in m.hypers])
    out["inc"], out["inc_val"] = m.get_incumbent()
    for nm, cls in (("ei", R.EI), ("log_ei", R.LogEI), ("pi", R.PI), ("lcb", R.LCB)):
        out["marg_" + nm] = R.MarginalizationGPMCMC(cls(m)).compute(inp["Xc"])
    # second BO iteration: burned, walkers continue from p0, RNG streams continue
    m.train(np.concatenate((inp["X"], inp["X2"])), np.concatenate((inp["y"], inp["y2"])), do_optimize=True)
    out["hypers2"] = np.array(m.hypers)
    out["mix_m2"], out["mix_v2"] = m.predict(inp["Xc"])
    # do_optimize=False: single model at the kernel's current vector + raw log-noise -8 (:144-147)
    m2 = R.GaussianProcessMCMC(2 * R.george.kernels.Matern52Kernel(np.ones(D), ndim=D), lower=inp["lower"],
                               upper=inp["upper"], rng=np.random.RandomState(3))
    m2.train(inp["X"], inp["y"], do_optimize=False)
    out["noopt_hypers"] = np.array(m2.hypers)
    out["noopt_m"], out["noopt_v"] = m2.predict(inp["Xc"])
    _save("ref_gpmcmc", **out)


# ------------------------------------------------------------------------------------------------
# (3) FabolasGP / FabolasGPMCMC (+ marginalised EI over it)
# ------------------------------------------------------------------------------------------------
def _fabolas_models(R, inp, which="obj"):
    D = inp["D"]
    basis = (lambda x: (1 - x) ** 2) if which == "obj" else (lambda x: x)
    target = inp["y"] if which == "obj" else inp["cost"]
    thetas = inp["thetas"] if which == "obj" else inp["thetas_cost"]
    mc = R.FabolasGPMCMC(george_fabolas_kernel(R, D), basis_func=basis, n_hypers=len(thetas), lower=inp["lower"],
                         upper=inp["upper"], rng=np.random.RandomState(5))
    mc.hypers = [t for t in thetas]
    mc.train(inp["X"], target, do_optimize=False)
    return mc, basis, target


def make_fabolas(R):
    inp = fabolas_inputs()
    D = inp["D"]
    th = inp["thetas"][0]
    gp = R.FabolasGP(george_fabolas_kernel(R, D, th[:-1]), basis_function=lambda x: (1 - x) ** 2,
                     noise=np.exp(th[-1]), lower=inp["lower"], upper=inp["upper"], rng=np.random.RandomState(1))
    gp.train(inp["X"], inp["y"], do_optimize=False)
    out = dict()
    out["mu"], out["var"] = gp.predict(inp["Xc"])
    out["inc"], out["inc_val"] = gp.get_incumbent()
    _, out["cov17"] = gp.predict(inp["Xc"][:17], full_cov=True)
    out["ei"] = R.EI(gp).compute(inp["Xc"])
    # last: the reference's nll() leaves theta in the SHARED kernel object (gaussian_process.py:151), i.e. it
    # changes what a later train(do_optimize=False) fits -- not mirrored, so nothing may follow it here
    out["nll"] = gp.nll(inp["thetas"][1])
    mc, _, _ = _fabolas_models(R, inp)
    out["mix_m"], out["mix_v"] = mc.predict(inp["Xc"])
    out["mcmc_inc"], out["mcmc_inc_val"] = mc.get_incumbent()
    for nm, cls in (("ei", R.EI), ("log_ei", R.LogEI), ("lcb", R.LCB)):
        out["marg_" + nm] = R.MarginalizationGPMCMC(cls(mc)).compute(inp["Xc"])
    _save("ref_fabolas", **out)


# ------------------------------------------------------------------------------------------------
# (4) InformationGain / InformationGainPerUnitCost, every candidate through the reference's loop
# ------------------------------------------------------------------------------------------------
def _ig_state(ig):
    return dict(zb=np.array(ig.zb), lmb=np.array(ig.lmb), logP=ig.logP, dlogPdMu=ig.dlogPdMu,
                dlogPdSigma=ig.dlogPdSigma, dlogPdMudMu=ig.dlogPdMudMu, sn2=ig.sn2)


def make_infogain(R):
    inp = infogain_inputs()
    D = inp["X"].shape[1]
    gp = R.GaussianProcess(george_kernel(R, "matern52", D, inp["theta"][:-1]), noise=np.exp(inp["theta"][-1]),
                           lower=inp["lower"], upper=inp["upper"], rng=np.random.RandomState(1))
    gp.train(inp["X"], inp["y"], do_optimize=False)
    np.random.seed(7)
    ig = R.InformationGain(gp, inp["lower"], inp["upper"], Nb=50, Np=400, rng=np.random.RandomState(8))
    ig.update(gp)
    out = _ig_state(ig)
    out["ig"] = ig.compute(inp["Xc"])
    out["argmax"] = int(np.argmax(out["ig"]))
    _save("ref_infogain", **out)

    # per unit cost on Fabolas models, marginalised over S hyper-parameter samples as fmin/fabolas.py:190-198 does
    finp = fabolas_inputs(M=120)
    D = finp["D"]
    mc_obj, _, _ = _fabolas_models(R, finp, "obj")
    mc_cost, _, _ = _fabolas_models(R, finp, "cost")
    lower, upper = np.append(finp["lower"], 0), np.append(finp["upper"], 1)
    is_env = np.zeros(D + 1)
    is_env[-1] = 1
    np.random.seed(9)
    igc = R.InformationGainPerUnitCost(mc_obj, mc_cost, lower, upper, sampling_acquisition=R.EI,
                                       is_env_variable=is_env, n_representer=20)
    marg = R.MarginalizationGPMCMC(igc)
    # NOTE marginalization.py:41-45 points every estimator's *model* at the COST sub-model when a cost model
    # exists; update() (called by the solver before every maximisation, fabolas.py:245) repairs that.
    marg.update(mc_obj, mc_cost, overhead=0.35)
    out = dict(S=len(marg.estimators), overhead=0.35)
    for i, e in enumerate(marg.estimators):
        for k, v in _ig_state(e).items():
            out["%s_%d" % (k, i)] = v
        out["ig_%d" % i] = e.compute(finp["Xc"])
        out["log_cost_%d" % i] = mc_cost.models[i].predict(finp["Xc"])[0]
    out["marg"] = marg.compute(finp["Xc"])
    _save("ref_infogain_cost", **out)


def make_infogain_config4(R, N=2048, D=10, M=256):
    """BASELINE config 4's shape (Fabolas kernel, D = 10 + 1, Nb = 50, Np = 400) at half its N through the
    reference's own per-candidate loop: representer sampling costs 2500+ single-point posteriors and every
    candidate a 51-RHS N x N solve, ~15 min of CPU at N = 2048 (N = 4096 did not finish in 50 min here)."""
    inp = fabolas_inputs(N=N, D=D, M=M, S=1, seed=71)
    th = inp["thetas"][0]
    gp = R.FabolasGP(george_fabolas_kernel(R, D, th[:-1]), basis_function=lambda x: (1 - x) ** 2,
                     noise=np.exp(th[-1]), lower=inp["lower"], upper=inp["upper"], rng=np.random.RandomState(1))
    gp.train(inp["X"], inp["y"], do_optimize=False)
    lower, upper = np.append(inp["lower"], 0), np.append(inp["upper"], 1)
    np.random.seed(11)
    ig = R.InformationGain(gp, lower, upper, Nb=50, Np=400, sampling_acquisition=R.EI,
                           rng=np.random.RandomState(12))
    ig.update(gp)
    out = _ig_state(ig)
    out["ig"] = ig.compute(inp["Xc"])
    _save("ref_infogain_config4", **out)


# ------------------------------------------------------------------------------------------------
# (5) BASELINE config 1: Branin through robo.fmin.bayesian_optimization, 30 iterations
# ------------------------------------------------------------------------------------------------
def _placeholder_optional_models():
    """robo/fmin/__init__.py imports every front end, and those import the optional model back ends
    (pybnn: DNGO/Bohamiann, pyrfr: random forests) at module level.  None of them is on the GP path or
    installed here; empty placeholder modules let ``robo.fmin`` import, any use would raise."""
    import types
    for mod, names in (("pybnn", ()), ("pybnn.dngo", ("DNGO",)), ("pybnn.bohamiann", ("Bohamiann",)),
                       ("pybnn.multi_task_bohamiann", ("MultiTaskBohamiann",)), ("pyrfr", ()),
                       ("pyrfr.regression", ())):
        if mod not in sys.modules:
            m = types.ModuleType(mod)
            for n in names:
                setattr(m, n, None)
            sys.modules[mod] = m


def make_branin(R, seed=3, n_iter=30):
    _placeholder_optional_models()
    from robo.fmin import bayesian_optimization as fmin_bo
    GP = R.GaussianProcess
    log = []
    orig_train = GP.train

    def train(self, X, y, do_optimize=True):
        orig_train(self, X, y, do_optimize)
        st = np.random.get_state()
        log.append(dict(n=X.shape[0], hypers=np.array(self.hypers, dtype=np.float64), noise=float(self.noise),
                        keys=st[1].copy(), pos=st[2], has_gauss=st[3], cached=st[4]))

    GP.train = train
    try:
        lo, hi = np.array([-5.0, 0.0]), np.array([10.0, 15.0])
        np.random.seed(seed)
        res = fmin_bo(branin, lo, hi, num_iterations=n_iter, n_init=3, model_type="gp", acquisition_func="ei",
                      maximizer="random", rng=np.random.RandomState(seed))
    finally:
        GP.train = orig_train
    out = dict(X=np.array(res["X"]), y=np.array(res["y"]), x_opt=np.array(res["x_opt"]), f_opt=res["f_opt"],
               incumbent_values=np.array(res["incumbent_values"]), seed=seed,
               n=np.array([l["n"] for l in log]), hypers=np.array([l["hypers"] for l in log]),
               noise=np.array([l["noise"] for l in log]), rng_keys=np.array([l["keys"] for l in log]),
               rng_pos=np.array([l["pos"] for l in log]), rng_has_gauss=np.array([l["has_gauss"] for l in log]),
               rng_cached=np.array([l["cached"] for l in log]))
    print("branin f_opt", res["f_opt"], "regret", res["f_opt"] - 0.397887)
    _save("ref_branin", **out)


# ------------------------------------------------------------------------------------------------
# (6) the other two front ends: robo.fmin.entropy_search (model="gp") and robo.fmin.fabolas, first
#     model-based iterations, everything their loops decide logged for a replay (tests/ref_checks.py)
# ------------------------------------------------------------------------------------------------
def _rng_state():
    st = np.random.get_state()
    return dict(keys=st[1].copy(), pos=st[2], has_gauss=st[3], cached=st[4])


def es_objective(x):
    """a 2-d objective on [-5, 10] x [0, 15] (Branin / 50: the scale entropy_search's default priors expect)"""
    return branin(x) / 50.0


def make_entropy_search(R, seed=5, n_iter=9):
    """robo/fmin/entropy_search.py:20-131 with model="gp": GaussianProcess (DefaultPrior, L-BFGS-B) +
    InformationGain(sampling_acquisition=EI, Nb=50, Np=400) + RandomSampling(500), run by the reference's own
    BayesianOptimization loop.  Logged per model-based iteration: the hyper-parameters its optimiser found, the
    representer points its (OS-entropy seeded) emcee sampler drew, the global RNG state right before
    RandomSampling.maximize, and what it chose."""
    _placeholder_optional_models()
    from robo.fmin import entropy_search as fmin_es
    from robo.maximizers.random_sampling import RandomSampling
    GP, IG = R.GaussianProcess, R.InformationGain
    log_t, log_r, log_m = [], [], []
    o_train, o_rep, o_max = GP.train, IG.sample_representer_points, RandomSampling.maximize

    def train(self, X, y, do_optimize=True):
        o_train(self, X, y, do_optimize)
        log_t.append(dict(n=X.shape[0], hypers=np.array(self.hypers, dtype=np.float64), noise=float(self.noise)))

    def rep(self):
        o_rep(self)
        log_r.append(dict(zb=np.array(self.zb), lmb=np.array(self.lmb)))

    def maximize(self):
        st = _rng_state()
        x = o_max(self)
        st["x"] = np.array(x)
        log_m.append(st)
        return x

    GP.train, IG.sample_representer_points, RandomSampling.maximize = train, rep, maximize
    try:
        lo, hi = np.array([-5.0, 0.0]), np.array([10.0, 15.0])
        np.random.seed(seed)
        res = fmin_es(es_objective, lo, hi, num_iterations=n_iter, n_init=3, maximizer="random", model="gp",
                      rng=np.random.RandomState(seed))
    finally:
        GP.train, IG.sample_representer_points, RandomSampling.maximize = o_train, o_rep, o_max
    assert len(log_t) == len(log_r) == len(log_m) == n_iter - 3
    _save("ref_entropy_search", X=np.array(res["X"]), y=np.array(res["y"]), seed=seed,
          n=np.array([l["n"] for l in log_t]), hypers=np.array([l["hypers"] for l in log_t]),
          noise=np.array([l["noise"] for l in log_t]), zb=np.array([l["zb"] for l in log_r]),
          lmb=np.array([l["lmb"] for l in log_r]), x_new=np.array([l["x"] for l in log_m]),
          rng_keys=np.array([l["keys"] for l in log_m]), rng_pos=np.array([l["pos"] for l in log_m]),
          rng_has_gauss=np.array([l["has_gauss"] for l in log_m]), rng_cached=np.array([l["cached"] for l in log_m]),
          incumbents=np.array(res["incumbents"]), incumbent_values=np.array(res["incumbent_values"]))


def fabolas_objective(x, s):
    """(validation error, cost) of a configuration x in [0,1]^2 on s data points: error falls, cost grows with s"""
    err = 0.1 + (x[0] - 0.3) ** 2 + 0.5 * (x[1] - 0.6) ** 2 + 2.0 / np.sqrt(s)
    return float(err), float(0.05 + s / 400.0 * (1.0 + x[0]))


def make_fabolas_frontend(R, seed=6, n_init=3, subsets=(64, 16), n_model_based=3):
    """robo/fmin/fabolas.py:31-312 as shipped: two FabolasGPMCMC models (EnvPrior, emcee), InformationGainPerUnitCost
    (EI proposal, 50 representer points) marginalised over the hyper-parameter samples, RandomSampling(500) on the
    extended box, projected incumbent.  Logged per model-based iteration: both models' hyper-parameter samples,
    every estimator's representer points, the incumbent estimate, the global RNG state before maximize, the choice."""
    _placeholder_optional_models()
    import robo.fmin                                  # noqa: F401
    ref_fab = sys.modules["robo.fmin.fabolas"]        # (the package attribute of that name is the function)
    from robo.maximizers.random_sampling import RandomSampling
    MC, IGC = R.FabolasGPMCMC, R.InformationGainPerUnitCost
    log_t, log_r, log_m, log_i = [], [], [], []
    o_train, o_rep, o_max, o_inc = MC.train, IGC.sample_representer_points, RandomSampling.maximize, \
        ref_fab.projected_incumbent_estimation

    def train(self, X, y, do_optimize=True, **kw):
        o_train(self, X, y, do_optimize, **kw)
        log_t.append(dict(n=X.shape[0], hypers=np.array(self.hypers, dtype=np.float64)))

    def rep(self):
        o_rep(self)
        log_r.append(dict(zb=np.array(self.zb), lmb=np.array(self.lmb)))

    def maximize(self):
        st = _rng_state()
        x = o_max(self)
        st["x"] = np.array(x)
        log_m.append(st)
        return x

    def inc(model, X, proj_value=1):
        a, b = o_inc(model, X, proj_value)
        log_i.append((np.array(a), float(b)))
        return a, b

    MC.train, IGC.sample_representer_points, RandomSampling.maximize = train, rep, maximize
    ref_fab.projected_incumbent_estimation = inc
    n0 = n_init * len(subsets)
    try:
        np.random.seed(seed)
        res = ref_fab.fabolas(fabolas_objective, np.zeros(2), np.ones(2), s_min=10, s_max=1000, n_init=n_init,
                              num_iterations=n0 + n_model_based, subsets=list(subsets), burnin=20, chain_length=10,
                              n_hypers=12, rng=np.random.RandomState(seed))
    finally:
        MC.train, IGC.sample_representer_points, RandomSampling.maximize = o_train, o_rep, o_max
        ref_fab.projected_incumbent_estimation = o_inc
    S = log_t[0]["hypers"].shape[0]
    assert len(log_t) == 2 * n_model_based + 1 and len(log_m) == n_model_based and len(log_r) == S * n_model_based
    out = dict(X=np.array(res["X"]), y=np.log(np.array(res["y"])), c=np.array(res["c"]), seed=seed, S=S, n0=n0,
               x_opt=np.array(res["x_opt"]))
    for it in range(n_model_based):
        out["hypers_obj_%d" % it] = log_t[2 * it]["hypers"]
        out["hypers_cost_%d" % it] = log_t[2 * it + 1]["hypers"]
        out["zb_%d" % it] = np.array([l["zb"] for l in log_r[S * it:S * (it + 1)]])
        out["lmb_%d" % it] = np.array([l["lmb"] for l in log_r[S * it:S * (it + 1)]])
        out["inc_%d" % it], out["inc_val_%d" % it] = log_i[it]
        m = log_m[it]
        out["x_new_%d" % it] = m["x"]
        out["rng_keys_%d" % it], out["rng_pos_%d" % it] = m["keys"], m["pos"]
        out["rng_has_gauss_%d" % it], out["rng_cached_%d" % it] = m["has_gauss"], m["cached"]
    out["hypers_final"] = log_t[-1]["hypers"]
    out["inc_final"], out["inc_val_final"] = log_i[-1]
    _save("ref_fabolas_frontend", **out)


# ------------------------------------------------------------------------------------------------
# (7) the reference's single-point maximisers behind its front end: robo.fmin.bayesian_optimization(maximizer="scipy" /
#     "differential_evolution") on Branin (robo/fmin/bayesian_optimization.py:131-139, robo/maximizers/scipy_optimizer.py,
#     differential_evolution.py), everything the loop decides logged for a replay (tests/ref_checks.py)
# ------------------------------------------------------------------------------------------------
def make_branin_single_point(R, seed=4, n_iter=11):
    _placeholder_optional_models()
    from robo.fmin import bayesian_optimization as fmin_bo
    GP = R.GaussianProcess
    orig_train = GP.train
    lo, hi = np.array([-5.0, 0.0]), np.array([10.0, 15.0])
    out = {"seed": seed}
    for name in ("scipy", "differential_evolution"):
        log = []

        def train(self, X, y, do_optimize=True):
            orig_train(self, X, y, do_optimize)
            st = np.random.get_state()
            log.append(dict(n=X.shape[0], hypers=np.array(self.hypers, dtype=np.float64), noise=float(self.noise),
                            keys=st[1].copy(), pos=st[2], has_gauss=st[3], cached=st[4]))

        GP.train = train
        try:
            np.random.seed(seed)
            res = fmin_bo(branin, lo, hi, num_iterations=n_iter, n_init=3, model_type="gp", acquisition_func="ei",
                          maximizer=name, rng=np.random.RandomState(seed))
        finally:
            GP.train = orig_train
        out.update({name + "_X": np.array(res["X"]), name + "_y": np.array(res["y"]), name + "_f_opt": res["f_opt"],
                    name + "_n": np.array([l["n"] for l in log]), name + "_hypers": np.array([l["hypers"] for l in log]),
                    name + "_noise": np.array([l["noise"] for l in log]),
                    name + "_rng_keys": np.array([l["keys"] for l in log]), name + "_rng_pos": np.array([l["pos"] for l in log]),
                    name + "_rng_has_gauss": np.array([l["has_gauss"] for l in log]),
                    name + "_rng_cached": np.array([l["cached"] for l in log])})
        print("branin", name, "f_opt", res["f_opt"], "model-based iterations", len(log))
    _save("ref_branin_single_point", **out)


# ------------------------------------------------------------------------------------------------
# (8) robo.fmin.bayesian_optimization(model_type="gp_mcmc", acquisition_func="log_ei"): the front end's own
#     MCMC configuration (DefaultPrior, 3 * len(kernel) walkers, 100 burn-in + 200 steps per iteration),
#     LogEI marginalised over the walkers' last positions (MarginalizationGPMCMC), RandomSampling
# ------------------------------------------------------------------------------------------------
def make_branin_gpmcmc(R, seed=7, n_iter=11):
    _placeholder_optional_models()
    from robo.fmin import bayesian_optimization as fmin_bo
    M = R.GaussianProcessMCMC
    log = []
    orig_train = M.train

    def train(self, X, y, do_optimize=True, **kw):
        before = self.rng.get_state()
        orig_train(self, X, y, do_optimize, **kw)
        st = np.random.get_state()
        after = self.rng.get_state()
        log.append(dict(n=X.shape[0], hypers=np.array(self.hypers, dtype=np.float64), keys=st[1].copy(), pos=st[2],
                        has_gauss=st[3], cached=st[4], own_keys=after[1].copy(), own_pos=after[2],
                        own_before_keys=before[1].copy(), own_before_pos=before[2]))

    M.train = train
    try:
        lo, hi = np.array([-5.0, 0.0]), np.array([10.0, 15.0])
        np.random.seed(seed)
        res = fmin_bo(branin, lo, hi, num_iterations=n_iter, n_init=3, model_type="gp_mcmc", acquisition_func="log_ei",
                      maximizer="random", rng=np.random.RandomState(seed))
    finally:
        M.train = orig_train
    out = dict(X=np.array(res["X"]), y=np.array(res["y"]), x_opt=np.array(res["x_opt"]), f_opt=res["f_opt"],
               incumbent_values=np.array(res["incumbent_values"]), seed=seed,
               n=np.array([l["n"] for l in log]), hypers=np.array([l["hypers"] for l in log]),
               rng_keys=np.array([l["keys"] for l in log]), rng_pos=np.array([l["pos"] for l in log]),
               rng_has_gauss=np.array([l["has_gauss"] for l in log]), rng_cached=np.array([l["cached"] for l in log]),
               own_keys=np.array([l["own_keys"] for l in log]), own_pos=np.array([l["own_pos"] for l in log]),
               own_before_keys=np.array([l["own_before_keys"] for l in log]),
               own_before_pos=np.array([l["own_before_pos"] for l in log]))
    print("branin gp_mcmc f_opt", res["f_opt"], "walkers", out["hypers"].shape)
    _save("ref_branin_gpmcmc", **out)


def make_branin_gpmcmc_acq(R, n_iter=8):
    """the same front end with the other three acquisition functions under MarginalizationGPMCMC: results only (the free
    run of robo_amd's front end with the same seeds must return them, tests/ref_checks.py)"""
    _placeholder_optional_models()
    from robo.fmin import bayesian_optimization as fmin_bo
    lo, hi = np.array([-5.0, 0.0]), np.array([10.0, 15.0])
    out = {}
    for seed, name in ((8, "ei"), (9, "pi"), (10, "lcb")):
        np.random.seed(seed)
        res = fmin_bo(branin, lo, hi, num_iterations=n_iter, n_init=3, model_type="gp_mcmc", acquisition_func=name,
                      maximizer="random", rng=np.random.RandomState(seed))
        out.update({name + "_seed": seed, name + "_X": np.array(res["X"]), name + "_y": np.array(res["y"]),
                    name + "_incumbent_values": np.array(res["incumbent_values"]), name + "_x_opt": np.array(res["x_opt"]),
                    name + "_f_opt": res["f_opt"]})
        print("branin gp_mcmc", name, "f_opt", res["f_opt"])
    _save("ref_branin_gpmcmc_acq", **out)


def make_entropy_search_gpmcmc(R, seed=12, n_iter=6):
    """robo/fmin/entropy_search.py with its DEFAULT model, "gp_mcmc": GaussianProcessMCMC (10 walkers, 100 + 200 steps) +
    MarginalizationGPMCMC(InformationGain(EI, Nb=50, Np=400)) + RandomSampling(500).  Logged per model-based iteration:
    the walkers' last positions, EVERY estimator's representer points (its emcee sampler is seeded from OS entropy in the
    reference: inputs of a replay), the global RNG state right before RandomSampling.maximize, and what it chose."""
    _placeholder_optional_models()
    from robo.fmin import entropy_search as fmin_es
    from robo.maximizers.random_sampling import RandomSampling
    M, IG = R.GaussianProcessMCMC, R.InformationGain
    log_t, log_r, log_m = [], [], []
    o_train, o_rep, o_max = M.train, IG.sample_representer_points, RandomSampling.maximize

    def train(self, X, y, do_optimize=True, **kw):
        o_train(self, X, y, do_optimize, **kw)
        log_t.append(dict(n=X.shape[0], hypers=np.array(self.hypers, dtype=np.float64)))

    def rep(self):
        o_rep(self)
        log_r.append(dict(zb=np.array(self.zb), lmb=np.array(self.lmb)))

    def maximize(self):
        st = _rng_state()
        x = o_max(self)
        st["x"] = np.array(x)
        log_m.append(st)
        return x

    M.train, IG.sample_representer_points, RandomSampling.maximize = train, rep, maximize
    try:
        lo, hi = np.array([-5.0, 0.0]), np.array([10.0, 15.0])
        np.random.seed(seed)
        res = fmin_es(es_objective, lo, hi, num_iterations=n_iter, n_init=3, maximizer="random", model="gp_mcmc",
                      rng=np.random.RandomState(seed))
    finally:
        M.train, IG.sample_representer_points, RandomSampling.maximize = o_train, o_rep, o_max
    n_mb = n_iter - 3
    S = log_t[0]["hypers"].shape[0]
    assert len(log_t) == len(log_m) == n_mb and len(log_r) == n_mb * S, (len(log_t), len(log_m), len(log_r), S)
    zb = np.array([l["zb"] for l in log_r]).reshape((n_mb, S) + log_r[0]["zb"].shape)
    lmb = np.array([l["lmb"] for l in log_r]).reshape((n_mb, S) + log_r[0]["lmb"].shape)
    _save("ref_entropy_search_gpmcmc", X=np.array(res["X"]), y=np.array(res["y"]), seed=seed,
          n=np.array([l["n"] for l in log_t]), hypers=np.array([l["hypers"] for l in log_t]), zb=zb, lmb=lmb,
          x_new=np.array([l["x"] for l in log_m]),
          rng_keys=np.array([l["keys"] for l in log_m]), rng_pos=np.array([l["pos"] for l in log_m]),
          rng_has_gauss=np.array([l["has_gauss"] for l in log_m]), rng_cached=np.array([l["cached"] for l in log_m]),
          incumbents=np.array(res["incumbents"]), incumbent_values=np.array(res["incumbent_values"]))


# ------------------------------------------------------------------------------------------------
# (9) the public surface on the path: names, parameter order and defaults of the reference's classes / functions,
#     and which public methods each class has (tests/test_api_surface.py holds robo_amd against it)
# ------------------------------------------------------------------------------------------------
API_SURFACE = [
    ("robo.models.gaussian_process", "GaussianProcess"), ("robo.models.gaussian_process_mcmc", "GaussianProcessMCMC"),
    ("robo.models.fabolas_gp", "FabolasGP"), ("robo.models.fabolas_gp", "FabolasGPMCMC"),
    ("robo.models.base_model", "BaseModel"),
    ("robo.acquisition_functions.ei", "EI"), ("robo.acquisition_functions.log_ei", "LogEI"),
    ("robo.acquisition_functions.pi", "PI"), ("robo.acquisition_functions.lcb", "LCB"),
    ("robo.acquisition_functions.information_gain", "InformationGain"),
    ("robo.acquisition_functions.information_gain_per_unit_cost", "InformationGainPerUnitCost"),
    ("robo.acquisition_functions.marginalization", "MarginalizationGPMCMC"),
    ("robo.acquisition_functions.base_acquisition", "BaseAcquisitionFunction"),
    ("robo.maximizers.random_sampling", "RandomSampling"), ("robo.maximizers.scipy_optimizer", "SciPyOptimizer"),
    ("robo.maximizers.differential_evolution", "DifferentialEvolution"), ("robo.maximizers.base_maximizer", "BaseMaximizer"),
    ("robo.maximizers.grid_search", "GridSearch"),
    ("robo.solver.bayesian_optimization", "BayesianOptimization"), ("robo.solver.base_solver", "BaseSolver"),
    ("robo.priors.default_priors", "DefaultPrior"), ("robo.priors.env_priors", "EnvPrior"),
    ("robo.priors.base_prior", "BasePrior"), ("robo.priors.base_prior", "TophatPrior"),
    ("robo.priors.base_prior", "HorseshoePrior"), ("robo.priors.base_prior", "LognormalPrior"),
    ("robo.priors.base_prior", "NormalPrior"),
    ("robo.fmin.bayesian_optimization", "bayesian_optimization"), ("robo.fmin.entropy_search", "entropy_search"),
    ("robo.fmin.fabolas", "fabolas"), ("robo.fmin.random_search", "random_search"),
    ("robo.initial_design.init_random_uniform", "init_random_uniform"),
    ("robo.initial_design.init_latin_hypercube_sampling", "init_latin_hypercube_sampling"),
    ("robo.initial_design.init_grid", "init_grid"), ("robo.initial_design.init_random_normal", "init_random_normal"),
    ("robo.util.incumbent_estimation", "projected_incumbent_estimation"),
    ("robo.util.normalization", "zero_one_normalization"), ("robo.util.normalization", "zero_one_unnormalization"),
    ("robo.util.normalization", "zero_mean_unit_var_normalization"),
    ("robo.util.normalization", "zero_mean_unit_var_unnormalization"),
    ("robo.util.epmgp", "joint_min"), ("robo.util.mc_part", "joint_pmin"),
]


def signature_of(f):
    """[[name, kind, default or None], ...]; callables as defaults by name, everything else by repr"""
    import inspect
    try:
        params = inspect.signature(f).parameters.values()
    except (TypeError, ValueError):
        return None
    out = []
    for p in params:
        if p.default is inspect.Parameter.empty:
            d = None
        elif callable(p.default):
            d = "<callable>:" + getattr(p.default, "__name__", "?")
        else:
            d = repr(p.default)
        out.append([p.name, p.kind.name, d])
    return out


def api_attribute_objects(ns, kernel_of, lo, hi, X, y):
    """one constructed-and-used instance per class, built the same way for the reference (ns = reference()) and for
    robo_amd (ns = its namespace; tests/test_api_surface.py) -> {class name: object}"""
    rs = np.random.RandomState
    objs = {}
    gp = ns.GaussianProcess(kernel_of(), prior=ns.DefaultPrior(4, rng=rs(1)), rng=rs(2), normalize_input=True, lower=lo,
                            upper=hi)
    gp.train(X, y, do_optimize=False)
    objs["GaussianProcess"] = gp
    mc = ns.GaussianProcessMCMC(kernel_of(), prior=ns.DefaultPrior(4, rng=rs(1)), n_hypers=8, chain_length=5,
                                burnin_steps=5, rng=rs(2), lower=lo, upper=hi)
    mc.train(X, y, do_optimize=True)
    objs["GaussianProcessMCMC"] = mc
    for name in ("EI", "LogEI", "PI", "LCB"):
        a = getattr(ns, name)(gp)
        a.update(gp)
        objs[name] = a
    ig = ns.InformationGain(gp, lo, hi, Nb=6, Np=10)
    ig.update(gp)
    objs["InformationGain"] = ig
    ma = ns.MarginalizationGPMCMC(ns.LogEI(mc))
    ma.update(mc)
    objs["MarginalizationGPMCMC"] = ma
    for name in ("RandomSampling", "SciPyOptimizer", "DifferentialEvolution"):
        objs[name] = getattr(ns, name)(ns.EI(gp), lo, hi, rng=rs(0))
    objs["BayesianOptimization"] = ns.BayesianOptimization(lambda x: float(np.sum(x)), lo, hi, ns.EI(gp), gp,
                                                           ns.RandomSampling(ns.EI(gp), lo, hi), rng=rs(0))
    objs["DefaultPrior"] = ns.DefaultPrior(4, rng=rs(0))
    objs["EnvPrior"] = ns.EnvPrior(5, n_ls=2, n_lr=2, rng=rs(0))
    return objs


def api_attribute_data():
    rs = np.random.RandomState(0)
    X = rs.rand(12, 2)
    return np.zeros(2), np.ones(2), X, np.sin(3 * X.sum(axis=1))


def _public_attributes(R):
    """public instance attributes of the reference's objects after construction + train / update"""
    from robo.maximizers.random_sampling import RandomSampling
    from robo.maximizers.scipy_optimizer import SciPyOptimizer
    from robo.maximizers.differential_evolution import DifferentialEvolution
    from robo.solver.bayesian_optimization import BayesianOptimization
    R.RandomSampling, R.SciPyOptimizer, R.DifferentialEvolution = RandomSampling, SciPyOptimizer, DifferentialEvolution
    R.BayesianOptimization = BayesianOptimization
    lo, hi, X, y = api_attribute_data()
    objs = api_attribute_objects(R, lambda: george_kernel(R, "matern52", 2, np.log([2.0, 1.0, 1.0])), lo, hi, X, y)
    return {k: sorted(a for a in vars(o) if not a.startswith("_")) for k, o in objs.items()}


def make_api_surface(R):
    import importlib
    import inspect
    import json
    _placeholder_optional_models()
    out = {}
    for mod, name in API_SURFACE:
        obj = getattr(importlib.import_module(mod), name)
        if inspect.isclass(obj):
            entry = {"__init__": signature_of(obj.__init__)}
            for m, member in inspect.getmembers(obj, predicate=inspect.isfunction):
                if not m.startswith("_"):
                    entry[m] = signature_of(member)
            out[mod + ":" + name] = entry
        else:
            out[mod + ":" + name] = {"": signature_of(obj)}
    n_obj, n_call = len(out), sum(len(v) for v in out.values())
    out["attributes"] = _public_attributes(R)
    with open(os.path.join(HERE, "ref_api_surface.json"), "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=True)
    print("wrote ref_api_surface.json:", n_obj, "objects,", n_call, "callables,", len(out["attributes"]), "attribute sets")


MAKERS = dict(gp=make_gp, mcmc=make_mcmc, fabolas=make_fabolas, infogain=make_infogain,
              infogain_config4=make_infogain_config4, branin=make_branin,
              entropy_search=make_entropy_search, fabolas_frontend=make_fabolas_frontend,
              branin_single_point=make_branin_single_point, branin_gpmcmc=make_branin_gpmcmc, branin_gpmcmc_acq=make_branin_gpmcmc_acq,
              entropy_search_gpmcmc=make_entropy_search_gpmcmc, api_surface=make_api_surface)

if __name__ == "__main__":
    R = reference()
    for nm in (sys.argv[1:] or list(MAKERS)):
        MAKERS[nm](R)

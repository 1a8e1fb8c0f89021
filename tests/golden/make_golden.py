#!/usr/bin/env python
"""Generate the golden fixtures under tests/golden/ (run in the BUILD container only).

    PYTHONPATH=/root/reference python tests/golden/make_golden.py

What is "real reference" and what is oracle here:

* The acquisition half is the REFERENCE'S OWN CODE: ``robo.acquisition_functions
  .{ei,log_ei,pi,lcb,marginalization}`` and ``robo.models.base_model`` import
  from /root/reference and are executed unchanged (``np.Infinity`` is aliased to
  ``np.inf`` first: the attribute no longer exists in NumPy 2 and the reference's
  degenerate LogEI branches use it, log_ei.py:89,96,118).
* The GP half (george) is not installable, so (mean, var) come from
  ``oracle.gp_oracle`` and are stored too; they are regression vectors for the
  oracle ("parity unpinned" for kernel values, see oracle/gp_oracle.py header).

Fixtures hold outputs + the seeds/shape parameters; tests regenerate the inputs
with ``golden_inputs`` below (imported by the tests) so files stay small.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import gp_oracle as O  # noqa: E402

CASES = {
    # name: kind, N, D, M, lower, upper, normalize_output, seed
    "small_matern": dict(kind="matern52", N=40, D=3, M=200, lo=-1.0, hi=2.0, nout=False, seed=11),
    "ragged_rbf_nout": dict(kind="rbf", N=67, D=5, M=257, lo=0.0, hi=1.0, nout=True, seed=12),
    "one_block_edge": dict(kind="matern52", N=127, D=2, M=129, lo=-5.0, hi=10.0, nout=False, seed=13),
    "two_block": dict(kind="matern52", N=200, D=6, M=300, lo=0.0, hi=1.0, nout=False, seed=14),
    "config2_sub": dict(kind="matern52", N=1024, D=8, M=8192, lo=0.0, hi=1.0, nout=False, seed=0),
}


def objective(X01):
    """The reference's test function, test/test_models/test_gaussian_process.py:15."""
    return np.sinc(X01 * 10 - 5).sum(axis=1)


def default_theta(kind, D):
    """log amp 0, log l^2 = log(0.25 D), log sigma^2 = log 1e-3."""
    return np.concatenate([[0.0], np.full(D, np.log(0.25 * D)), [np.log(1e-3)]])


def golden_inputs(case):
    c = CASES[case] if isinstance(case, str) else case
    rng = np.random.RandomState(c["seed"])
    X01 = rng.rand(c["N"], c["D"])
    y = objective(X01)
    y = (y - y.mean()) / y.std()
    lower = np.full(c["D"], c["lo"])
    upper = np.full(c["D"], c["hi"])
    X = lower + (upper - lower) * X01
    Xc = lower + (upper - lower) * np.random.RandomState(c["seed"] + 1000).rand(c["M"], c["D"])
    theta = default_theta(c["kind"], c["D"])
    # make theta non-isotropic so that an ARD index mix-up is visible
    theta[1:-1] += 0.2 * np.random.RandomState(c["seed"] + 2000).randn(c["D"])
    return dict(X=X, y=y, Xc=Xc, theta=theta, lower=lower, upper=upper, kind=c["kind"],
                nout=c["nout"])


def mcmc_inputs():
    rng = np.random.RandomState(21)
    N, D, M, S = 50, 4, 300, 6
    X = rng.rand(N, D)
    y = objective(X)
    Xc = np.random.RandomState(22).rand(M, D)
    base = default_theta("matern52", D)
    thetas = base[None, :] + 0.3 * np.random.RandomState(2).randn(S, base.size)
    return dict(X=X, y=y, Xc=Xc, thetas=thetas, lower=np.zeros(D), upper=np.ones(D), kind="matern52")


def _reference_modules():
    if not hasattr(np, "Infinity"):
        np.Infinity = np.inf      # NumPy-2 hazard in log_ei.py:89,96,118
    sys.path.insert(0, "/root/reference")
    from robo.models.base_model import BaseModel
    from robo.acquisition_functions.ei import EI
    from robo.acquisition_functions.log_ei import LogEI
    from robo.acquisition_functions.pi import PI
    from robo.acquisition_functions.lcb import LCB
    from robo.acquisition_functions.marginalization import MarginalizationGPMCMC
    return BaseModel, EI, LogEI, PI, LCB, MarginalizationGPMCMC


def main():
    BaseModel, EI, LogEI, PI, LCB, Marg = _reference_modules()

    class RefModel(BaseModel):
        """reference BaseModel whose predict is served by the oracle GP."""

        def __init__(self, gp):
            self.gp = gp
            self.X, self.y = gp.X, gp.y

        def train(self, X, y):
            pass

        def predict(self, X_test, **kw):
            return self.gp.predict(X_test, diag_only=True)

        def get_incumbent(self):
            return self.gp.get_incumbent()

    class FixedModel(BaseModel):
        """returns prescribed (m, v): drives the reference's degenerate branches."""

        def __init__(self, m, v, eta):
            self.m, self.v, self.eta = m, v, eta

        def train(self, X, y):
            pass

        def predict(self, X_test, **kw):
            return self.m.copy(), self.v.copy()

        def get_incumbent(self):
            return None, self.eta

    # ---- GP cases ------------------------------------------------------------
    for name in CASES:
        inp = golden_inputs(name)
        gp = O.OracleGP(inp["kind"], inp["theta"], normalize_output=inp["nout"],
                        lower=inp["lower"], upper=inp["upper"])
        gp.train(inp["X"], inp["y"])
        model = RefModel(gp)
        mu, var = gp.predict(inp["Xc"], diag_only=True)
        _, eta = gp.get_incumbent()
        out = dict(mu=mu, var=var, eta=eta,
                   loglik=gp.loglikelihood(inp["theta"]),
                   ei=EI(model).compute(inp["Xc"]),
                   log_ei=LogEI(model).compute(inp["Xc"]),
                   pi=PI(model).compute(inp["Xc"]),
                   lcb=LCB(model).compute(inp["Xc"]),
                   ei_par=EI(model, par=0.3).compute(inp["Xc"]),
                   lcb_par=LCB(model, par=2.5).compute(inp["Xc"]))
        if inp["X"].shape[0] <= 256:
            # the reference call sequence (full covariance then np.diag) on the same inputs
            mu_f, var_f = gp.predict(inp["Xc"])
            out["mu_fullcov_path"] = mu_f
            out["var_fullcov_path"] = var_f
            _, cov = gp.predict(inp["Xc"][:33], full_cov=True)
            out["cov33"] = cov
        for k in ("ei", "log_ei", "pi", "lcb"):
            out["argmax_" + k] = int(np.argmax(out[k]))
            srt = np.sort(out[k])
            out["gap_" + k] = float(srt[-1] - srt[-2])
        np.savez(os.path.join(HERE, name + ".npz"), **out)
        print(name, "ei max", out["ei"].max(), "argmax", out["argmax_ei"], "gap", out["gap_ei"])

    # ---- GP-MCMC marginalisation ------------------------------------------------
    inp = mcmc_inputs()
    gps = []
    for th in inp["thetas"]:
        g = O.OracleGP(inp["kind"], th, lower=inp["lower"], upper=inp["upper"])
        g.train(inp["X"], inp["y"])
        gps.append(g)

    class RefMCMC(BaseModel):
        def __init__(self, models):
            self.models = models
            self.X, self.y = models[0].X, models[0].y

        def train(self, X, y):
            pass

        def predict(self, X_test, **kw):
            mu = np.array([m.predict(X_test)[0] for m in self.models])
            var = np.array([m.predict(X_test)[1] for m in self.models])
            return O.mcmc_mixture(mu, var)

        def get_incumbent(self):
            return self.models[0].get_incumbent()

    mm = RefMCMC([RefModel(g) for g in gps])
    out = dict()
    for nm, cls in (("ei", EI), ("log_ei", LogEI), ("pi", PI), ("lcb", LCB)):
        out["marg_" + nm] = Marg(cls(mm)).compute(inp["Xc"])
    out["mu_s"] = np.array([g.predict(inp["Xc"], diag_only=True)[0] for g in gps])
    out["var_s"] = np.array([g.predict(inp["Xc"], diag_only=True)[1] for g in gps])
    out["mix_m"], out["mix_v"] = O.mcmc_mixture(out["mu_s"], out["var_s"])
    out["loglik_s"] = np.array([g.loglikelihood(g.theta) for g in gps])
    np.savez(os.path.join(HERE, "mcmc_marginal.npz"), **out)
    print("mcmc", out["marg_log_ei"][:3])

    # ---- element-wise acquisition sweep incl. degenerate branches -------------------
    rng = np.random.RandomState(5)
    eta = 0.25
    m = np.concatenate([rng.randn(400) * 2, [eta, eta, eta - 1.0, eta + 1.0, eta + 30.0, eta - 30.0,
                                             eta + 1e-9, eta - 1e-9, eta + 5.0, eta + 5.0]])
    v = np.concatenate([np.exp(rng.randn(400) * 3), [1.0, 0.0, 0.0, 0.0, 1e-4, 1e-4,
                                                    1e-30, 1e-30, 2.22e-16, 1e-3]])
    # vectors for PI/LCB/LogEI may contain sigma == 0; EI collapses the whole batch then, so
    # EI gets the strictly-positive-variance subset.
    fm = FixedModel(m, v, eta)
    dummyX = np.zeros((m.size, 1))
    with np.errstate(all="ignore"):
        out = dict(m=m, v=v, eta=eta,
                   log_ei=LogEI(fm).compute(dummyX),
                   log_ei_par=LogEI(fm, par=0.1).compute(dummyX),
                   pi=PI(fm).compute(dummyX),
                   lcb=LCB(fm).compute(dummyX))
    pos = v > 0
    fm2 = FixedModel(m[pos], v[pos], eta)
    out["pos"] = pos
    out["ei_pos"] = EI(fm2).compute(dummyX[pos])
    out["ei_collapsed"] = np.asarray(EI(fm).compute(dummyX), dtype=np.float64)
    np.savez(os.path.join(HERE, "acq_elementwise.npz"), **out)

    # ---- the reference's DemoModel pins (test/dummy_model.py:6-21) --------
    sys.path.insert(0, "/root/reference/test")
    from dummy_model import DemoModel
    rs = np.random.RandomState(0)
    X = rs.rand(10, 2)
    y = np.sinc(X * 10 - 5).sum(axis=1)
    dm = DemoModel()
    dm.train(X, y)
    Xt = rs.rand(5, 2)
    pins = dict(X=X, y=y, Xt=Xt,
                ei=EI(dm).compute(Xt), log_ei=LogEI(dm).compute(Xt),
                pi=PI(dm).compute(Xt), lcb=LCB(dm).compute(Xt))
    np.savez(os.path.join(HERE, "demo_model_pins.npz"), **pins)
    print("demo pins", pins["ei"][0], pins["log_ei"][0], pins["pi"][0], pins["lcb"][0])


if __name__ == "__main__":
    main()

"""Host-side pieces that sit next to the hot path: kernel descriptions, priors, initial
designs, candidate recipe, ensemble sampler, sharding helpers.  Where the reference's own
module is importable (/root/reference present, i.e. the build container) outputs are compared
with it on seeded inputs; on the GPU box those comparisons are skipped."""
import os
import sys

import numpy as np
import pytest

from robo_amd import sharding
from robo_amd.initial_design import init_latin_hypercube_sampling, init_random_uniform
from robo_amd.kernels import ExpSquaredKernel, Matern52Kernel
from robo_amd.priors import DefaultPrior, HorseshoePrior, LognormalPrior, NormalPrior, TophatPrior
from robo_amd.util.ensemble_sampler import EnsembleSampler

HAVE_REF = os.path.isdir("/root/reference/robo")
needs_ref = pytest.mark.skipif(not HAVE_REF, reason="reference tree not on this box")


def _ref():
    if "/root/reference" not in sys.path:
        sys.path.insert(0, "/root/reference")


def test_kernel_api_slice():
    k = 2 * Matern52Kernel(np.ones(3), ndim=3)
    assert len(k) == 4 and k.kind == "matern52"
    np.testing.assert_allclose(k.get_parameter_vector(), [np.log(2.0 / 3), 0, 0, 0])
    k.set_parameter_vector(np.array([0.1, 0.2, 0.3, 0.4]))
    np.testing.assert_array_equal(k[:], [0.1, 0.2, 0.3, 0.4])
    assert isinstance(k[:].tolist(), list)
    assert ExpSquaredKernel(0.5, ndim=2).kind == "rbf"
    import copy
    k2 = copy.deepcopy(k)
    k2.set_parameter_vector(np.zeros(4))
    assert k[0] == 0.1


@needs_ref
def test_priors_match_reference():
    _ref()
    from robo.priors import base_prior as rb
    from robo.priors.default_priors import DefaultPrior as RefDefault
    th = np.array([0.3, -1.2, 0.7, 1.9, -4.0])
    for mine, ref in ((TophatPrior(-10, 2), rb.TophatPrior(-10, 2)),
                      (HorseshoePrior(0.1), rb.HorseshoePrior(0.1)),
                      (LognormalPrior(1.0, 0.0), rb.LognormalPrior(1.0, 0.0)),
                      (NormalPrior(2.0, 0.5), rb.NormalPrior(2.0, 0.5))):
        np.testing.assert_array_equal(np.asarray(mine.lnprob(th[1:3])), np.asarray(ref.lnprob(th[1:3])))
    assert DefaultPrior(5).lnprob(th) == RefDefault(5).lnprob(th)
    assert DefaultPrior(5).lnprob(np.array([0.3, 3.0, 0.7, 1.9, -4.0])) == -np.inf
    a = DefaultPrior(5, rng=np.random.RandomState(7)).sample_from_prior(6)
    b = RefDefault(5, rng=np.random.RandomState(7)).sample_from_prior(6)
    np.testing.assert_array_equal(a, b)


@needs_ref
def test_initial_designs_match_reference_draw_order():
    _ref()
    from robo.initial_design import init_latin_hypercube_sampling as ref_lhs
    from robo.initial_design import init_random_uniform as ref_uni
    lo, hi = np.array([-5.0, 0.0, 1.0]), np.array([10.0, 15.0, 2.0])
    np.testing.assert_array_equal(init_random_uniform(lo, hi, 9, rng=np.random.RandomState(3)),
                                  ref_uni(lo, hi, 9, rng=np.random.RandomState(3)))
    np.testing.assert_array_equal(init_latin_hypercube_sampling(lo, hi, 7, rng=np.random.RandomState(3)),
                                  ref_lhs(lo, hi, 7, rng=np.random.RandomState(3)))


def test_default_prior_batch_equals_scalar():
    pr = DefaultPrior(6, rng=np.random.RandomState(0))
    th = np.random.RandomState(1).randn(40, 6) * 3
    th[3, 2] = 5.0          # outside the tophat
    th[7, -1] = 0.0         # the horseshoe's +inf point
    with np.errstate(all="ignore"):
        want = np.array([pr.lnprob(t) for t in th])
        got = pr.lnprob_batch(th)
    fin = np.isfinite(want)
    np.testing.assert_array_equal(got[~fin], want[~fin])          # same -inf / +inf pattern
    np.testing.assert_allclose(got[fin], want[fin], rtol=1e-13, atol=1e-13)   # closed-form lognormal vs scipy


def test_initial_designs_properties():
    lo, hi = np.array([-5.0, 0.0]), np.array([10.0, 15.0])
    P = init_latin_hypercube_sampling(lo, hi, 10, rng=np.random.RandomState(0))
    assert P.shape == (10, 2)
    for d in range(2):   # exactly one point per stratum
        strata = np.floor((P[:, d] - lo[d]) / (hi[d] - lo[d]) * 10).astype(int)
        assert sorted(strata) == list(range(10))
    U = init_random_uniform(lo, hi, 50, rng=np.random.RandomState(0))
    assert np.all(U >= lo) and np.all(U <= hi)


def test_grid_and_normal_designs():
    """robo/initial_design/{init_grid,init_random_normal}.py: pinned values (rows in np.meshgrid 'xy' order; a normal draw of
    n numbers per dimension, dimension by dimension, clipped) and, where the reference tree is present, its own output"""
    from robo_amd.initial_design import init_grid, init_random_normal
    lo, hi = np.array([-1.0, 0.0, 2.0]), np.array([1.0, 5.0, 3.0])
    g = init_grid(lo[:2], hi[:2], 3)
    np.testing.assert_array_equal(g, [[-1, 0], [0, 0], [1, 0], [-1, 2.5], [0, 2.5], [1, 2.5], [-1, 5], [0, 5], [1, 5]])
    assert init_grid(lo, hi, 2).shape == (8, 3)
    r = np.random.RandomState(3)
    want = np.stack([np.clip(r.normal(m, 0.1, 6), a, b) for m, a, b in zip(0.5 * (lo + hi), lo, hi)], axis=1)
    np.testing.assert_array_equal(init_random_normal(lo, hi, 6, rng=np.random.RandomState(3)), want)
    wide = init_random_normal(lo, hi, 200, std=np.full(3, 50.0), rng=np.random.RandomState(4))
    assert np.all(wide >= lo) and np.all(wide <= hi) and np.any(wide == lo) and np.any(wide == hi)      # clipped
    if os.path.isdir("/root/reference/robo"):
        _ref()
        from robo.initial_design import init_grid as ref_grid, init_random_normal as ref_normal
        for n in (1, 2, 4):
            np.testing.assert_array_equal(init_grid(lo, hi, n), ref_grid(lo, hi, n))
        for kw in ({}, {"mean": np.array([0.0, 4.9, 2.1])}, {"std": np.array([1.0, 2.0, 3.0])}):
            np.testing.assert_array_equal(init_random_normal(lo, hi, 7, rng=np.random.RandomState(3), **kw),
                                          ref_normal(lo, hi, 7, rng=np.random.RandomState(3), **kw))


@needs_ref
@pytest.mark.parametrize("D", [2, 3])
def test_random_sampling_candidates_match_reference(D):
    """same candidate batch as robo/maximizers/random_sampling.py:38-47 under the same global seed (the reference draws
    row by row in Python loops; here one block call per recipe part -- same stream, same values, also for odd D where a
    Gaussian draw leaves a cached second value behind)"""
    _ref()
    from robo.maximizers.random_sampling import RandomSampling as RefRS
    from robo_amd.maximizers import RandomSampling

    class Acq(object):
        class model(object):
            @staticmethod
            def get_incumbent():
                return np.array([0.3, 0.6, 0.95][:D]), 0.0

        def __init__(self):
            self.seen = None

        def __call__(self, X):
            self.seen = X
            return -np.sum((X - 0.4) ** 2, axis=1)

    lo, hi = np.zeros(D), np.array([1.0, 2.0, 1.5][:D])
    a, b = Acq(), Acq()
    np.random.seed(11)
    x_ref = RefRS(b, lo, hi, n_samples=201).maximize()
    np.random.seed(11)
    x = RandomSampling(a, lo, hi, n_samples=201).maximize()
    np.testing.assert_array_equal(a.seen, b.seen)
    np.testing.assert_array_equal(x, x_ref)


def test_ensemble_sampler_recovers_gaussian():
    mean = np.array([1.0, -2.0, 0.5])
    std = np.array([0.5, 2.0, 1.0])
    calls = []

    def lnp_batch(th):
        calls.append(th.shape[0])
        return -0.5 * np.sum(((th - mean) / std) ** 2, axis=1)

    s = EnsembleSampler(20, 3, lnprob_batch=lnp_batch)
    rng = np.random.RandomState(0)
    p0 = rng.randn(20, 3)
    p, lnp, _ = s.run_mcmc(p0, 300, rstate0=rng)
    p, lnp, _ = s.run_mcmc(p, 1500, rstate0=rng)
    ch = s.chain[:, 300:, :].reshape(-1, 3)
    assert np.all(np.abs(ch.mean(axis=0) - mean) < 0.15 * std + 0.05)
    assert np.all(np.abs(ch.std(axis=0) / std - 1) < 0.15)
    assert s.chain.shape == (20, 1800, 3) and set(calls) == {10, 20}   # initial lnprob of all walkers, then half-ensemble batches
    assert 0.2 < s.acceptance_fraction.mean() < 0.9
    # -inf proposals are rejected, walkers stay finite
    s2 = EnsembleSampler(8, 2, lnprob_batch=lambda th: np.where(th[:, 0] > 0, -np.sum(th ** 2, axis=1), -np.inf))
    p, lnp, _ = s2.run_mcmc(np.abs(rng.randn(8, 2)) + 0.1, 200, rstate0=rng)
    assert np.all(p[:, 0] > 0) and np.all(np.isfinite(lnp))


def test_shard_ranges():
    assert [sharding.shard_range(50, r, 4) for r in range(4)] == [(0, 13), (13, 26), (26, 38), (38, 50)]
    assert [sharding.shard_range(8, r, 8) for r in range(8)] == [(i, i + 1) for i in range(8)]
    spans = [sharding.shard_range(65536 * 8, r, 8) for r in range(8)]
    assert spans[0] == (0, 65536) and spans[-1][1] == 65536 * 8


def test_reduce_argmax_matches_numpy():
    rs = np.random.RandomState(0)
    for trial in range(50):
        y = rs.randint(0, 5, size=40).astype(float)
        if trial % 3 == 0:
            y[rs.randint(40)] = np.nan
        pairs = []
        for r in range(4):
            b, e = sharding.shard_range(40, r, 4)
            j = int(np.argmax(y[b:e]))
            pairs.append((y[b + j], b + j))
        v, i = sharding.reduce_argmax(reversed(pairs))     # arrival order must not matter
        assert i == int(np.argmax(y))


def test_marginalization_without_opt_in_never_touches_the_communicator(monkeypatch):
    """an initialised process group alone must not make MarginalizationGPMCMC create a communicator (a
    collective): the opt-in flag is looked at before sharding.dist_info()"""
    from robo_amd.acquisition_functions.marginalization import MarginalizationGPMCMC

    def boom():
        raise AssertionError("dist_info() called although sample_shard is off")

    monkeypatch.setattr(sharding, "dist_info", boom)
    acq = MarginalizationGPMCMC.__new__(MarginalizationGPMCMC)
    acq.sample_shard = False
    acq.estimators = []
    assert acq._shard() is None


def test_base_solver_helpers(tmp_path):
    """robo/solver/base_solver.py:41-139: run directory, observations, the per-iteration JSON record"""
    import json
    from robo_amd.solver import BaseSolver, BayesianOptimization

    class Part(object):
        def __init__(self, tag):
            self.tag = tag

        def get_json_data(self):
            return {"tag": self.tag}

    assert issubclass(BayesianOptimization, BaseSolver)
    s = BaseSolver(acquisition_func=Part("a"), model=Part("m"), maximize_func=None, task=Part("t"),
                   save_dir=str(tmp_path / "run"))
    s.create_save_dir()                                   # an existing directory is reused
    assert s.get_model().tag == "m" and BaseSolver().get_model() is None
    s.X, s.y = np.zeros((2, 1)), np.ones(2)
    assert s.get_observations()[1] is s.y
    s.time_overhead, s.time_func_eval, s.time_start = [0.5], [0.25], 0.0
    s.incumbent, s.incumbent_value = np.array([0.1]), np.array([2.0])
    rec = s.get_json_data(0)
    assert rec["iteration"] == 0 and rec["incumbent"] == [0.1] and rec["incumbent_fval"] == [2.0] and \
        rec["optimization_overhead"] == 0.5 and rec["time_func_eval"] == 0.25 and rec["runtime"] > 0
    s.save_json(0)
    s.output_file_json.close()
    s.output_file.close()
    line = json.loads(open(str(tmp_path / "run" / "results.json")).read().strip())
    assert sorted(line) == ["Acquisiton", "Model", "Solver", "Task"] and line["Model"] == {"tag": "m"}


def test_compat_import_alias():
    """robo_amd.compat: ``robo.x.y`` is the robo_amd module of the same path (same objects), ``george.kernels`` the kernel
    module; out-of-scope modules raise ImportError; uninstall leaves nothing behind"""
    import importlib
    import robo_amd.compat as compat
    saved = {k: v for k, v in sys.modules.items() if k in ("robo", "george") or k.startswith(("robo.", "george."))}
    path = list(sys.path)
    sys.path[:] = [p for p in sys.path if os.path.abspath(p or ".") != "/root/reference"]      # the real package out of reach
    try:
        compat.uninstall()
        compat.install(force=True)
        compat.install()                                           # idempotent
        import robo_amd.fmin
        import robo_amd.models.gaussian_process
        assert importlib.import_module("robo.fmin").bayesian_optimization is robo_amd.fmin.bayesian_optimization
        assert importlib.import_module("robo.models.gaussian_process") is robo_amd.models.gaussian_process
        # the aliased modules keep their OWN import attributes (importlib would stamp the alias spec on them: reload
        # would become a no-op through the alias loader, runpy / pkgutil / inspect would see the wrong name)
        assert robo_amd.fmin.__spec__.name == "robo_amd.fmin" and robo_amd.fmin.__name__ == "robo_amd.fmin"
        assert robo_amd.models.gaussian_process.__spec__.name == "robo_amd.models.gaussian_process"
        assert type(robo_amd.models.gaussian_process.__spec__.loader).__name__ != "_RoboAlias"
        for path_ in ("robo.priors.default_priors", "robo.priors.env_priors", "robo.priors.base_prior",
                      "robo.initial_design.init_grid", "robo.maximizers.base_maximizer", "robo.solver.base_solver",
                      "robo.util.mc_part", "robo.acquisition_functions.marginalization"):
            importlib.import_module(path_)
        george = importlib.import_module("george")
        k = 2 * george.kernels.Matern52Kernel(np.ones(3), ndim=3)
        assert len(k) == 4 and len(george.kernels.Matern52Kernel(np.ones(3), ndim=3)) == 3
        with pytest.raises(ImportError):
            importlib.import_module("robo.models.random_forest")
        assert not hasattr(george, "GP")
        compat.uninstall()
        if os.path.isdir("/root/reference/robo"):
            sys.path.insert(0, "/root/reference")                  # the real package within reach: refuse unless forced
            with pytest.raises(RuntimeError):
                compat.install()
            sys.path.remove("/root/reference")
    finally:
        compat.uninstall()
        sys.path[:] = path
        assert "robo" not in sys.modules and "george" not in sys.modules
        sys.modules.update(saved)


def test_random_search_front_end(tmp_path):
    """robo/fmin/random_search.py: result keys, incumbent = best so far, per-iteration JSON, no state carried between calls;
    where the reference tree is present: its points and incumbents under the same seed (also its (1, 1) points in 1-D)"""
    import json
    from robo_amd.fmin import random_search
    f = lambda x: float(np.sum((np.asarray(x) - 0.3) ** 2))      # noqa: E731
    lo, hi = np.zeros(2), np.ones(2)
    r = random_search(f, lo, hi, num_iterations=9, rng=np.random.RandomState(1), output_path=str(tmp_path))
    assert sorted(r) == ["X", "f_opt", "incumbent_values", "incumbents", "overhead", "runtime", "time_func_eval", "x_opt", "y"]
    assert len(r["X"]) == 9 and r["f_opt"] == min(r["y"]) and r["x_opt"] == r["X"][int(np.argmin(r["y"]))]
    assert all(a >= b for a, b in zip(r["incumbent_values"], r["incumbent_values"][1:]))
    assert json.load(open(str(tmp_path / "robo_iter_8.json")))["iteration"] == 8
    again = random_search(f, lo, hi, num_iterations=9, rng=np.random.RandomState(1))
    assert again["X"] == r["X"] and len(again["X"]) == 9
    if os.path.isdir("/root/reference/robo"):
        _ref()
        import importlib.util
        spec = importlib.util.spec_from_file_location("_ref_random_search", "/root/reference/robo/fmin/random_search.py")
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        for lo_, hi_ in ((lo, hi), (np.array([0.0]), np.array([6.0]))):
            a = mod.random_search(f, lo_, hi_, X_init=[], Y_init=[], num_iterations=7, rng=np.random.RandomState(4))
            b = random_search(f, lo_, hi_, num_iterations=7, rng=np.random.RandomState(4))
            for k in ("X", "y", "x_opt", "f_opt", "incumbents", "incumbent_values"):
                assert a[k] == b[k], k


def test_fabolas_kernel_in_george_product_notation():
    """robo/fmin/fabolas.py:104-117 multiplies the Fabolas kernel up from single-axis george kernels; the same lines on
    robo_amd.kernels give the FabolasKernel the front end builds directly (same parameter vector, george's product order)"""
    import copy
    import pickle
    import robo_amd.kernels as K
    for D, amp in ((2, 1), (4, 3.0)):
        kernel = amp
        for d in range(D):
            kernel *= K.Matern52Kernel(np.ones([1]) * 0.01, ndim=D + 1, axes=d)
        kernel *= K.BayesianLinearRegressionKernel(log_a=0.1, log_b=0.1, ndim=D + 1, axes=D)
        assert isinstance(kernel, K.FabolasKernel) and len(kernel) == 1 + D + 2 and kernel.kind == "fabolas"
        np.testing.assert_array_equal(kernel.get_parameter_vector(), K.FabolasKernel(D + 1, amp=amp).get_parameter_vector())
    part = 1 * K.Matern52Kernel(np.ones([1]) * 0.01, ndim=4, axes=0)
    with pytest.raises(NotImplementedError):
        len(part)                                       # an incomplete product has no device kernel
    with pytest.raises(NotImplementedError):
        K.ExpSquaredKernel(np.ones(1), ndim=3, axes=1)
    k = 2.0 * K.Matern52Kernel(np.ones(3), ndim=3)
    assert len(pickle.loads(pickle.dumps(k))) == 4 and len(copy.deepcopy(k)) == 4 and type(k) is K.Matern52Kernel

"""End-to-end drop-in: the REFERENCE's own BayesianOptimization + RandomSampling
(robo/solver/bayesian_optimization.py, robo/maximizers/random_sampling.py, imported unchanged
from /root/reference) drive

  (a) robo_amd's GaussianProcess + EI/LogEI/LCB  (HIP sources; here interpreted by tests/hipemu), and
  (b) the reference's own EI/LogEI/LCB on top of an oracle-backed reference BaseModel,

with identical seeds.  Both must choose the SAME candidate index at every iteration, i.e.
produce the same (X, y) trajectory -- the "argmax index identical" rule of the north star,
checked through the whole solver loop.  Needs the reference tree (build container only); the
GPU-box equivalent with robo_amd's own loop is tests/test_gpu_parity.py::test_host_classes_on_gpu.
"""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
pytestmark = pytest.mark.skipif(not os.path.isdir("/root/reference/robo"), reason="reference tree not on this box")


def branin(x):
    x1, x2 = x
    return (x2 - 5.1 * x1 ** 2 / (4 * np.pi ** 2) + 5 * x1 / np.pi - 6) ** 2 + \
        10 * (1 - 1 / (8 * np.pi)) * np.cos(x1) + 10


@pytest.fixture(scope="module")
def emu():
    sys.path.insert(0, os.path.join(HERE, "hipemu"))
    import build_emu
    from robo_amd import _lib
    _lib.use_library(build_emu.build())
    yield
    _lib.use_library(None)


def _reference():
    if not hasattr(np, "Infinity"):
        np.Infinity = np.inf
    if "/root/reference" not in sys.path:
        sys.path.insert(0, "/root/reference")
    from robo.solver.bayesian_optimization import BayesianOptimization
    from robo.maximizers.random_sampling import RandomSampling
    from robo.initial_design import init_latin_hypercube_sampling
    from robo.models.base_model import BaseModel
    from robo.acquisition_functions.ei import EI
    from robo.acquisition_functions.log_ei import LogEI
    from robo.acquisition_functions.lcb import LCB
    return BayesianOptimization, RandomSampling, init_latin_hypercube_sampling, BaseModel, \
        {"ei": EI, "log_ei": LogEI, "lcb": LCB}


THETA = np.array([np.log(2.0 / 2), np.log(0.3), np.log(0.3), np.log(1e-3)])   # fixed hypers (do_optimize=False)


def _run(model, acq, n_iter, seed):
    BO, RS, lhs, _, _ = _reference()
    lo, hi = np.array([-5.0, 0.0]), np.array([10.0, 15.0])
    rng = np.random.RandomState(seed)
    np.random.seed(seed)          # RandomSampling uses the GLOBAL rng (random_sampling.py:38-45)
    bo = BO(branin, lo, hi, acq, model, RS(acq, lo, hi, rng=rng), initial_design=lhs, initial_points=3, rng=rng,
            train_interval=10 ** 9)   # do_optimize only at it % interval == 0 -> never after the design
    bo.run(n_iter)
    return np.array(bo.X), np.array(bo.y)


@pytest.mark.parametrize("acq_name", ["ei", "log_ei", "lcb"])
def test_same_trajectory_as_reference_classes_on_the_oracle(emu, acq_name):
    from oracle import gp_oracle as O
    from robo_amd.kernels import Matern52Kernel
    from robo_amd.models import GaussianProcess
    from robo_amd import acquisition_functions as A
    BO, RS, lhs, BaseModel, ref_acq = _reference()
    lo, hi = np.array([-5.0, 0.0]), np.array([10.0, 15.0])

    class OracleModel(BaseModel):
        """reference BaseModel whose GP is the oracle (fixed hypers)"""

        def train(self, X, y, do_optimize=False):
            self.gp = O.OracleGP("matern52", THETA, lower=lo, upper=hi)
            self.gp.train(X, y)
            self.X, self.y = self.gp.X, self.gp.y

        def predict(self, X_test, **kw):
            return self.gp.predict(X_test, diag_only=True)

        def get_incumbent(self):
            return self.gp.get_incumbent()

    ref_model = OracleModel()
    Xr, yr = _run(ref_model, ref_acq[acq_name](ref_model), 12, seed=5)

    class FixedGP(GaussianProcess):
        def train(self, X, y, do_optimize=True):     # the solver's first iteration asks for optimisation
            super(FixedGP, self).train(X, y, do_optimize=False)

    kernel = Matern52Kernel(np.exp(THETA[1:-1]), ndim=2, log_amp=THETA[0])
    model = FixedGP(kernel, noise=np.exp(THETA[-1]), lower=lo, upper=hi, rng=np.random.RandomState(0))
    mine = {"ei": A.EI, "log_ei": A.LogEI, "lcb": A.LCB}[acq_name](model)
    Xm, ym = _run(model, mine, 12, seed=5)

    np.testing.assert_array_equal(Xm, Xr)      # identical candidate chosen at every iteration
    np.testing.assert_array_equal(ym, yr)


def test_reference_solver_bookkeeping_with_robo_amd_objects(emu):
    """the assertions of the reference's test/test_solver/test_bayesian_optimization.py:28-49"""
    from robo_amd.kernels import Matern52Kernel
    from robo_amd.models import GaussianProcess
    from robo_amd.acquisition_functions import EI
    BO, RS, lhs, _, _ = _reference()
    lo, hi = np.zeros(1), np.ones(1) * 6
    model = GaussianProcess(2 * Matern52Kernel(np.ones(1), ndim=1), lower=lo, upper=hi, rng=np.random.RandomState(1))
    acq = EI(model)
    bo = BO(lambda x: (x[0] - 2.2) ** 2, lo, hi, acq, model, RS(acq, lo, hi, rng=np.random.RandomState(1)),
            initial_points=2, rng=np.random.RandomState(1))
    inc, val = bo.run(4)
    for lst in (bo.time_overhead, bo.time_func_evals, bo.incumbents, bo.incumbents_values, bo.runtime):
        assert len(lst) == 4
    assert bo.X.shape == (4, 1) and bo.y.shape == (4,)
    x = bo.choose_next(bo.X, bo.y)
    assert x.shape == (1,) and lo[0] <= x[0] <= hi[0]


def test_generic_plugin_model_with_robo_amd_acquisitions(emu):
    """any BaseModel works: the reference's DemoModel (test/dummy_model.py) + robo_amd EI/LogEI/PI/LCB
    reproduce the pins of SURVEY.md 8(c)"""
    _reference()
    sys.path.insert(0, "/root/reference/test")
    from dummy_model import DemoModel
    from robo_amd import acquisition_functions as A
    rs = np.random.RandomState(0)
    X = rs.rand(10, 2)
    y = np.sinc(X * 10 - 5).sum(axis=1)
    dm = DemoModel()
    dm.train(X, y)
    Xt = rs.rand(5, 2)
    gold = np.load(os.path.join(HERE, "golden", "demo_model_pins.npz"))
    np.testing.assert_allclose(A.EI(dm).compute(Xt), gold["ei"], rtol=1e-12)
    np.testing.assert_allclose(A.LogEI(dm).compute(Xt), gold["log_ei"], rtol=1e-12)
    np.testing.assert_allclose(A.PI(dm).compute(Xt), gold["pi"], rtol=1e-12)
    np.testing.assert_allclose(A.LCB(dm).compute(Xt), gold["lcb"], rtol=1e-12)
    assert A.EI(dm).compute(Xt).shape == (5,)


def test_reference_single_point_maximizers_drive_robo_amd_acquisitions(emu):
    """The reference's OTHER maximisers -- GridSearch, SciPyOptimizer, DifferentialEvolution (robo/maximizers/*.py,
    imported unchanged) -- call the acquisition with ONE point at a time (scipy_optimizer.py:44,
    differential_evolution.py:29, grid_search.py:60; SURVEY 8b "who calls it").  Driving robo_amd's EI on the device GP
    they must land where they land with the reference's own EI on the oracle-backed reference model (same seeds): the
    same grid point; the same optimum of the restarts / the evolution to optimiser tolerance."""
    from oracle import gp_oracle as O
    from robo_amd.kernels import Matern52Kernel
    from robo_amd.models import GaussianProcess
    from robo_amd import acquisition_functions as A
    _, _, _, BaseModel, ref_acq = _reference()
    from robo.maximizers.grid_search import GridSearch
    from robo.maximizers.scipy_optimizer import SciPyOptimizer
    from robo.maximizers.differential_evolution import DifferentialEvolution

    def models(lo, hi, X, y, theta):
        class OracleModel(BaseModel):
            def train(self, X, y, do_optimize=False):
                self.gp = O.OracleGP("matern52", theta, lower=lo, upper=hi)
                self.gp.train(X, y)
                self.X, self.y = self.gp.X, self.gp.y

            def predict(self, X_test, **kw):
                return self.gp.predict(X_test, diag_only=True)

            def get_incumbent(self):
                return self.gp.get_incumbent()

        ref_model = OracleModel()
        ref_model.train(X, y)
        D = lo.shape[0]
        mine = GaussianProcess(Matern52Kernel(np.exp(theta[1:-1]), ndim=D, log_amp=theta[0]), noise=np.exp(theta[-1]),
                               lower=lo, upper=hi, rng=np.random.RandomState(0))
        mine.train(X, y, do_optimize=False)
        return ref_acq["ei"](ref_model), A.EI(mine)

    # 1-d: the grid search picks the same grid point
    lo, hi = np.zeros(1), np.ones(1) * 6
    rs = np.random.RandomState(3)
    X = rs.rand(9, 1) * 6
    y = (X[:, 0] - 2.2) ** 2
    ref_ei, my_ei = models(lo, hi, X, y, np.array([np.log(4.0), np.log(0.1), np.log(1e-3)]))
    np.testing.assert_array_equal(GridSearch(my_ei, lo, hi, resolution=200).maximize(),
                                  GridSearch(ref_ei, lo, hi, resolution=200).maximize())
    # 2-d Branin: L-BFGS-B restarts and differential evolution, global NumPy stream seeded as the reference uses it
    lo, hi = np.array([-5.0, 0.0]), np.array([10.0, 15.0])
    X = lo + (hi - lo) * rs.rand(14, 2)
    y = np.array([branin(x) for x in X])
    ref_ei, my_ei = models(lo, hi, X, y, THETA)
    for make in (lambda acq: SciPyOptimizer(acq, lo, hi, n_restarts=6, rng=np.random.RandomState(4)),
                 lambda acq: DifferentialEvolution(acq, lo, hi, n_iters=8, rng=np.random.RandomState(4))):
        np.random.seed(9)
        x_ref = make(ref_ei).maximize()
        np.random.seed(9)
        x_mine = make(my_ei).maximize()
        assert np.all(x_mine >= lo) and np.all(x_mine <= hi)
        np.testing.assert_allclose(x_mine, x_ref, rtol=0, atol=1e-3 * (hi - lo).max())
        np.testing.assert_allclose(my_ei(x_mine[None, :]), ref_ei(x_ref[None, :]), rtol=1e-5, atol=1e-12)


def test_models_survive_deepcopy_and_pickle_and_the_reference_marginalisation(emu):
    """the reference deep-copies acquisition functions together with their models
    (robo/acquisition_functions/marginalization.py:36,67), so an object holding a ctypes handle must survive
    copy.deepcopy / pickle.  A trained GaussianProcess and a trained GaussianProcessMCMC are copied both ways and predict
    the same numbers; the REFERENCE's own MarginalizationGPMCMC (imported unchanged) wraps robo_amd's EI over robo_amd's
    GaussianProcessMCMC and returns what robo_amd's fused marginalisation returns."""
    import copy
    import pickle
    from robo_amd.kernels import Matern52Kernel
    from robo_amd.models import GaussianProcess, GaussianProcessMCMC
    from robo_amd.priors import DefaultPrior
    from robo_amd import acquisition_functions as A
    _reference()
    from robo.acquisition_functions.marginalization import MarginalizationGPMCMC as RefMarginalization
    lo, hi = np.array([-5.0, 0.0]), np.array([10.0, 15.0])
    rs = np.random.RandomState(2)
    X = lo + (hi - lo) * rs.rand(16, 2)
    y = np.array([branin(x) for x in X])
    Xt = lo + (hi - lo) * rs.rand(33, 2)
    gp = GaussianProcess(Matern52Kernel(np.exp(THETA[1:-1]), ndim=2, log_amp=THETA[0]), noise=np.exp(THETA[-1]), lower=lo,
                         upper=hi, rng=np.random.RandomState(0))
    gp.train(X, y, do_optimize=False)
    want = gp.predict(Xt)
    for clone in (copy.deepcopy(gp), pickle.loads(pickle.dumps(gp))):
        got = clone.predict(Xt)
        np.testing.assert_array_equal(got[0], want[0])
        np.testing.assert_array_equal(got[1], want[1])
        np.testing.assert_array_equal(A.EI(clone).compute(Xt), A.EI(gp).compute(Xt))
    kernel = 2 * Matern52Kernel(np.ones(2), ndim=2)
    mc = GaussianProcessMCMC(kernel, prior=DefaultPrior(len(kernel) + 1, rng=np.random.RandomState(3)), n_hypers=8,
                             chain_length=6, burnin_steps=6, lower=lo, upper=hi, rng=np.random.RandomState(4))
    mc.train(X, y)
    want = mc.predict(Xt)
    for clone in (copy.deepcopy(mc), pickle.loads(pickle.dumps(mc))):
        got = clone.predict(Xt)
        np.testing.assert_allclose(got[0], want[0], rtol=1e-13)
        np.testing.assert_allclose(got[1], want[1], rtol=1e-12)
    # the reference's marginalisation: deep copies of robo_amd's EI, one per hyper-parameter sample (marginalization.py:34-46)
    for cls in (A.EI, A.LogEI, A.LCB):
        ref_marg = RefMarginalization(cls(mc))
        ref_marg.update(mc)
        mine = A.MarginalizationGPMCMC(cls(mc))
        mine.update(mc)
        np.testing.assert_allclose(ref_marg.compute(Xt), mine.compute(Xt), rtol=1e-12, atol=1e-15)


def test_batched_differential_evolution(emu):
    """DifferentialEvolution(batched=True): SciPy's vectorised form, one acquisition call per generation instead of one per
    trial point -- the same optimum as the reference-shaped form, in a fraction of the calls"""
    from robo_amd.kernels import Matern52Kernel
    from robo_amd.models import GaussianProcess
    from robo_amd import acquisition_functions as A
    from robo_amd.maximizers import DifferentialEvolution
    rs = np.random.RandomState(0)
    lo, hi = np.zeros(2), np.ones(2)
    X = rs.rand(12, 2)
    y = np.sin(5 * X.sum(axis=1))
    gp = GaussianProcess(2 * Matern52Kernel(np.ones(2), ndim=2), lower=lo, upper=hi, rng=np.random.RandomState(1))
    gp.train(X, y, do_optimize=False)

    class Counting(A.EI):
        calls = 0

        def compute(self, X, **kw):
            Counting.calls += 1
            return super(Counting, self).compute(X, **kw)

    acq = Counting(gp)
    out = {}
    for batched in (False, True):
        Counting.calls = 0
        np.random.seed(3)
        out[batched] = (DifferentialEvolution(acq, lo, hi, n_iters=10, batched=batched).maximize(), Counting.calls)
    np.testing.assert_allclose(out[True][0], out[False][0], atol=1e-3)
    np.testing.assert_allclose(acq(out[True][0][None, :]), acq(out[False][0][None, :]), rtol=1e-5)
    assert out[True][1] * 5 < out[False][1], (out[True][1], out[False][1])

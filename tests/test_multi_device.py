"""Single-process multi-device entry points (robo_amd/csrc/multi.hip, include/robo_hip.h "multi-GPU, ONE process")
through the g++ interpreter build with HIPEMU_DEVICES emulated devices: 2 and 3 devices in ONE process (and two contexts
on one device) must give the single-device result for the candidate shard, the sample shard, the per-unit-cost shard,
the replicated / batched fits and the mixture posterior.  CPU-only; the GPU suite repeats the checks with two contexts
on the MI355X (tests/test_gpu_parity.py::test_multi_device_*)."""
import os
import sys

import numpy as np
import pytest

import multi_checks as MC
from robo_amd import _lib

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def emu3():
    sys.path.insert(0, os.path.join(HERE, "hipemu"))
    import build_emu
    path = build_emu.build()
    old = os.environ.get("HIPEMU_DEVICES")
    os.environ["HIPEMU_DEVICES"] = "3"
    _lib.use_library(path)
    assert _lib.device_count() == 3
    yield
    _lib.use_library(None)
    if old is None:
        del os.environ["HIPEMU_DEVICES"]
    else:
        os.environ["HIPEMU_DEVICES"] = old


@pytest.mark.parametrize("devices", [[0, 1], [0, 1, 2], [0, 0]])
def test_candidate_shard(emu3, devices):
    MC.check_candidate_shard(devices)


@pytest.mark.parametrize("devices", [[0, 1], [0, 1, 2]])
def test_sample_shard(emu3, devices):
    MC.check_sample_shard(devices)


@pytest.mark.parametrize("devices", [[0, 1], [0, 1, 2]])
def test_per_unit_cost_shard(emu3, devices):
    MC.check_per_cost_shard(devices)


def test_fits_and_mixture(emu3):
    MC.check_fits_and_mixture([0, 1, 2])


def test_failing_device_is_reported_not_hung(emu3):
    MC.check_failing_device([0, 1])


def test_same_results_without_worker_threads(emu3, monkeypatch):
    monkeypatch.setenv("ROBO_MULTI_THREADS", "0")
    _lib._multis.clear()
    try:
        assert _lib.multi_for([0, 1]).info()[2] == 0
        MC.check_candidate_shard([0, 1])
        MC.check_sample_shard([0, 1])
    finally:
        _lib._multis.clear()


def test_sample_shard_between_devices_that_are_no_peers(emu3, monkeypatch):
    """hipDeviceCanAccessPeer says no (HIPEMU_NO_PEER; on hardware ROBO_MULTI_NO_PEER=1 forces the same path): the partial
    sums and per-sample posteriors of a sample shard reach the first device through pinned host memory
    (multi.hip gather_to_first) -- same bits as the peer copies, same bits as one device"""
    monkeypatch.setenv("HIPEMU_NO_PEER", "1")
    _lib._multis.clear()
    try:
        MC.check_sample_shard([0, 1, 2])
        MC.check_fits_and_mixture([0, 1, 2])
    finally:
        _lib._multis.clear()
    monkeypatch.delenv("HIPEMU_NO_PEER")
    monkeypatch.setenv("ROBO_MULTI_NO_PEER", "1")       # the library's own switch, devices that WOULD be peers
    try:
        MC.check_sample_shard([0, 1])
    finally:
        _lib._multis.clear()


def test_device_resolution(emu3):
    assert _lib.resolve_devices(None, None) is None
    assert _lib.resolve_devices(None, 1) is None
    assert _lib.resolve_devices(None, 3) == [0, 1, 2]
    assert _lib.resolve_devices([2, 0]) == [2, 0]
    with pytest.raises(ValueError):
        _lib.resolve_devices([0, 5])
    m = _lib.multi_for([0, 1, 2])
    assert m.info()[:2] == (3, [0, 1, 2]) and m is _lib.multi_for((0, 1, 2))
    assert [_lib.shard_range(50, g, 4) for g in range(4)] == [(0, 13), (13, 26), (26, 38), (38, 50)]


# ---- the plugin classes and the robo.fmin front ends on several devices of one process -----------------------------------
def test_front_ends_same_trajectory_on_2_and_3_devices(emu3):
    """robo_amd.fmin.bayesian_optimization(n_gpus / devices): model_type="gp" splits the candidate batch of every
    maximisation over replicas on G devices of THIS process, "gp_mcmc" splits the hyper-parameter samples (fits, marginal
    LogEI, mixture posterior) -> the points the one-device run chooses, bit for bit; the objective is evaluated once per
    iteration (robo/solver/bayesian_optimization.py:156-203)"""
    MC.check_front_end_trajectories([[0, 1], [0, 1, 2]])
    XG, calls = MC.run_bo(n_gpus=3, model_type="gp", acquisition_func="lcb", maximizer="device_random", n_candidates=700)
    assert XG.shape == (8, 2) and calls == 8


def test_gp_mcmc_classes_on_devices(emu3):
    """GaussianProcessMCMC(devices=...) + MarginalizationGPMCMC: where the samples live, the batched fits per device, the
    mixture posterior and the marginal acquisition against the one-device objects; walker shard of the chain"""
    from robo_amd.acquisition_functions import LogEI, MarginalizationGPMCMC
    from robo_amd.kernels import Matern52Kernel
    from robo_amd.models import GaussianProcessMCMC
    from robo_amd.priors import DefaultPrior
    rs = np.random.RandomState(4)
    lo, hi = np.zeros(3), np.ones(3)
    X = rs.rand(40, 3)
    y = np.sin(3 * X.sum(axis=1))
    Xc = rs.rand(200, 3)

    def build(devices, walker_min=10 ** 9):
        kernel = 2 * Matern52Kernel(np.ones(3), ndim=3)
        m = GaussianProcessMCMC(kernel, prior=DefaultPrior(len(kernel) + 1, rng=np.random.RandomState(5)), n_hypers=10, chain_length=4, burnin_steps=6,
                                rng=np.random.RandomState(7), lower=lo, upper=hi, devices=devices)
        m.walker_shard_min_n = walker_min
        m.train(X, y)
        return m
    m1, m3 = build(None), build([0, 1, 2])
    np.testing.assert_array_equal(np.array(m3.hypers), np.array(m1.hypers))
    ctxs = _lib.multi_for([0, 1, 2]).ctxs
    assert [ctxs.index(s.gp.ctx) for s in m3.models] == [0, 0, 0, 0, 1, 1, 1, 2, 2, 2]
    mu1, v1 = m1.predict(Xc)
    mu3, v3 = m3.predict(Xc)
    np.testing.assert_array_equal(mu3, mu1)
    np.testing.assert_array_equal(v3, v1)
    a1, a3 = MarginalizationGPMCMC(LogEI(m1)), MarginalizationGPMCMC(LogEI(m3))
    a1.update(m1)
    a3.update(m3)
    np.testing.assert_allclose(a3.compute(Xc), a1.compute(Xc), rtol=1e-12)
    assert a3.argmax(Xc) == a1.argmax(Xc) == int(np.argmax(a1.compute(Xc)))
    # a pickled copy: ten samples whose handles rematerialise on the DEFAULT context at the first predict (the pickle drops
    # the sub-models' device slots); the second predict must see that and keep off the multi-device entry point
    import pickle
    clone = pickle.loads(pickle.dumps(m3))
    for rep in range(2):
        for a, b in zip(clone.predict(Xc), (mu1, v1)):
            np.testing.assert_allclose(a, b, rtol=1e-12, atol=1e-14)
        assert not clone._on_their_slots()
    # walkers split over the devices per half-step (host sampler around robo_gp_loglik_batch_multi): the same chain
    MC.check_walker_shard([0, 1, 2])


def test_information_gain_candidate_shard(emu3):
    """InformationGain / InformationGainPerUnitCost on GaussianProcess(devices=...): gains of a candidate batch split over
    the replicas == the one-device values for the SAME representer points and EP state"""
    from robo_amd.acquisition_functions import EI, InformationGain, InformationGainPerUnitCost
    from robo_amd.kernels import FabolasKernel, Matern52Kernel
    from robo_amd.models import FabolasGP, GaussianProcess
    rs = np.random.RandomState(8)
    lo, hi = np.zeros(2), np.ones(2)
    X = rs.rand(30, 2)
    y = np.sin(4 * X.sum(axis=1))
    Xc = rs.rand(101, 2)
    gains = []
    state = None
    for devices in (None, [0, 1, 2]):
        gp = GaussianProcess(2 * Matern52Kernel(np.ones(2), ndim=2), lower=lo, upper=hi, rng=np.random.RandomState(1),
                             devices=devices)
        gp.train(X, y, do_optimize=False)
        ig = InformationGain(gp, lo, hi, Nb=8, Np=20, sampling_acquisition=EI, rng=np.random.RandomState(2))
        ig.update(gp)
        if state is None:
            state = {k: getattr(ig, k) for k in ("zb", "lmb", "logP", "dlogPdMu", "dlogPdSigma", "dlogPdMudMu", "W", "_ep")}
        else:
            ig.__dict__.update(state)
        gains.append((ig.compute(Xc), ig.argmax(Xc)))
    np.testing.assert_array_equal(gains[1][0], gains[0][0])
    assert gains[1][1] == gains[0][1] == int(np.argmax(gains[0][0]))
    # per unit cost (Fabolas kernels, config 4's shape)
    lo3, hi3 = np.zeros(3), np.ones(3)
    Xf = rs.rand(36, 3)
    yf, cf = np.sin(3 * Xf.sum(axis=1)), 0.3 * Xf[:, -1] + 0.05 * rs.randn(36)
    Xcf = rs.rand(77, 3)
    out, state = [], None
    for devices in (None, [0, 1]):
        mk = lambda basis: FabolasGP(FabolasKernel(3, metric=0.3, log_a=0.1, log_b=0.1, amp=1.0), basis_function=basis,
                                     lower=lo3[:2], upper=hi3[:2], rng=np.random.RandomState(1), devices=devices)
        gm, cm = mk(lambda s: (1 - s) ** 2), mk(lambda s: s)
        gm.train(Xf, yf, do_optimize=False)
        cm.train(Xf, cf, do_optimize=False)
        ig = InformationGainPerUnitCost(gm, cm, lo3, hi3, is_env_variable=np.array([0, 0, 1]), sampling_acquisition=EI,
                                        n_representer=8, Np=20, rng=np.random.RandomState(2))
        ig.update(gm, cm, overhead=0.1)
        if state is None:
            state = {k: getattr(ig, k) for k in ("zb", "lmb", "logP", "dlogPdMu", "dlogPdSigma", "dlogPdMudMu", "W", "_ep")}
        else:
            ig.__dict__.update(state)
        out.append((ig.compute(Xcf), ig.argmax(Xcf)))
    np.testing.assert_array_equal(out[1][0], out[0][0])
    assert out[1][1] == out[0][1]


def test_fabolas_objects_on_devices(emu3):
    """what robo_amd.fmin.fabolas(n_gpus=2) wires together (build_fabolas(devices=[0, 1])), through one model-based iteration
    of its loop: both models' samples split over two devices, every sample's information gain per unit cost evaluated on
    its device by its own host thread; the threaded mean equals the sequential one"""
    import inspect
    from robo_amd.fmin.fabolas import build_fabolas, fabolas
    assert {"n_gpus", "devices"} <= set(inspect.signature(fabolas).parameters)
    lo, hi = np.zeros(1), np.ones(1)
    mo, mc, acq, maxi = build_fabolas(lo, hi, burnin=3, chain_length=2, rng=np.random.RandomState(3), n_candidates=40,
                                      n_representer=6, n_outcomes=10, devices=[0, 1])
    rs = np.random.RandomState(9)
    X = rs.rand(8, 2)
    y, c = np.log(np.exp(-np.sum((X[:, :1] - 0.3) ** 2, axis=1)) + 0.5 + 0.2 * (1 - X[:, 1])), np.log(0.1 + X[:, 1])
    mo.train(X, y)
    mc.train(X, c)
    acq.update(mo, mc)
    ctxs = _lib.multi_for([0, 1]).ctxs
    slots = [ctxs.index(e.model.gp.ctx) for e in acq.estimators]
    assert slots == sorted(slots) and set(slots) == {0, 1}
    assert [ctxs.index(e.cost_model.gp.ctx) for e in acq.estimators] == slots      # loss and cost sample s share a device
    Xt = rs.rand(30, 2)
    threaded = acq.compute(Xt)
    seq = np.mean([e.compute(Xt) for e in acq.estimators], axis=0)
    np.testing.assert_array_equal(threaded, seq)
    x_new = maxi.maximize()
    assert x_new.shape == (2,) and np.all(x_new >= 0) and np.all(x_new <= 1)


def test_more_devices_than_samples_and_copies(emu3):
    """edge cases of the device lists: fewer hyper-parameter samples than devices (a device without a sample), a model
    trained without hyper-parameter inference (ONE sample), fewer candidates than devices, and deepcopy / pickle of models
    that hold replicas (the copies re-create their device state on first use, on the same device list)"""
    import copy
    import pickle
    from robo_amd.acquisition_functions import EI, LogEI, MarginalizationGPMCMC
    from robo_amd.kernels import Matern52Kernel
    from robo_amd.models import GaussianProcess, GaussianProcessMCMC
    from robo_amd.priors import DefaultPrior
    rs = np.random.RandomState(12)
    lo, hi = np.zeros(2), np.ones(2)
    X = rs.rand(25, 2)
    y = np.cos(4 * X.sum(axis=1))
    Xc = rs.rand(57, 2)

    def mcmc(devices, n_hypers):
        kernel = 2 * Matern52Kernel(np.ones(2), ndim=2)
        return GaussianProcessMCMC(kernel, prior=DefaultPrior(len(kernel) + 1, rng=np.random.RandomState(5)), n_hypers=n_hypers,
                                   chain_length=3, burnin_steps=3, rng=np.random.RandomState(7), lower=lo, upper=hi,
                                   devices=devices)
    for n_hypers, optimise in ((8, True), (8, False)):       # 3 / 3 / 2 samples; ONE sample on three devices
        m1, m3 = mcmc(None, n_hypers), mcmc([0, 1, 2], n_hypers)
        m1.train(X, y, do_optimize=optimise)
        m3.train(X, y, do_optimize=optimise)
        assert len(m3.models) == (n_hypers if optimise else 1)
        for a, b in zip(m3.predict(Xc), m1.predict(Xc)):
            np.testing.assert_array_equal(a, b)
        a1, a3 = MarginalizationGPMCMC(LogEI(m1)), MarginalizationGPMCMC(LogEI(m3))
        a1.update(m1)
        a3.update(m3)
        np.testing.assert_allclose(a3.compute(Xc), a1.compute(Xc), rtol=1e-12)
        assert a3.argmax(Xc) == a1.argmax(Xc)
    # candidate shard with fewer candidates than devices, and copies of a model with replicas
    gp1 = GaussianProcess(2 * Matern52Kernel(np.ones(2), ndim=2), lower=lo, upper=hi, rng=np.random.RandomState(1))
    gp3 = GaussianProcess(2 * Matern52Kernel(np.ones(2), ndim=2), lower=lo, upper=hi, rng=np.random.RandomState(1),
                          devices=[0, 1, 2])
    gp1.train(X, y, do_optimize=False)
    gp3.train(X, y, do_optimize=False)
    assert len(gp3.replicas) == 2
    for n in (1, 2, 3, 57):
        np.testing.assert_array_equal(EI(gp3).compute(Xc[:n]), EI(gp1).compute(Xc[:n]))
        assert EI(gp3).argmax(Xc[:n]) == EI(gp1).argmax(Xc[:n])
    for clone in (copy.deepcopy(gp3), pickle.loads(pickle.dumps(gp3))):
        assert clone.devices == [0, 1, 2] and clone.gp is None and clone.replicas == []
        np.testing.assert_array_equal(EI(clone).compute(Xc), EI(gp1).compute(Xc))
        assert len(clone.replicas) == 2
    clone = pickle.loads(pickle.dumps(m3))
    for a, b in zip(clone.predict(Xc), m1.predict(Xc)):
        np.testing.assert_array_equal(a, b)
    # ... and AGAIN: the first call rematerialised the sub-models' handles on the default context (the pickle drops
    # their device slots); the second must not hand them to the multi-device entry point (round-5 advice: it did, and
    # failed with a context mismatch) but take the per-model path, same numbers to rounding
    for a, b in zip(clone.predict(Xc), m1.predict(Xc)):
        np.testing.assert_allclose(a, b, rtol=1e-12, atol=1e-14)


def test_reference_gp_mcmc_run_replayed_on_3_devices(emu3):
    """the REFERENCE'S own robo.fmin.bayesian_optimization(model_type="gp_mcmc") run (fixture ref_branin_gpmcmc): with the
    10 hyper-parameter samples spread over three devices of this process (4/3/3: batched fits per device, per-device
    partial sums added in device order) the marginal LogEI still picks the reference's candidate (first five model-based iterations here)"""
    import ref_checks as R
    checked, gap = R.check_ref_branin_gpmcmc_replay(devices=[0, 1, 2], chain=False, max_iters=5)
    assert checked == 5 and gap > 1e-7, (checked, gap)
    # and the public entry point itself: bayesian_optimization(model_type="gp_mcmc", n_gpus=3) with the reference's seeds
    # returns the REFERENCE'S run (first 5 of its 11 points here), not merely its own one-device run
    assert R.check_ref_branin_gpmcmc_free_run(num_iterations=5, n_gpus=3) == 5


def test_create_destroy_cycles_leak_nothing(emu3):
    """40 create / use / destroy cycles of the multi-device handle with random shapes (ragged shards, fewer candidates than
    devices, two contexts on one device): the candidate shard equals the single-device call every time, no context and
    no worker thread is left behind"""
    rng = np.random.default_rng(5)
    before = _lib.live_contexts()
    tasks = lambda: len(os.listdir("/proc/self/task"))      # noqa: E731
    n_tasks = None
    for cycle in range(40):
        devs = [[0, 1], [0, 1, 2], [0, 0], [2, 1]][cycle % 4]
        ctxs = [_lib.Context(d) for d in devs]
        multi = _lib.Multi(ctxs)
        N, D, M = int(rng.integers(5, 60)), int(rng.integers(1, 4)), int(rng.integers(1, 90))
        X, y = rng.random((N, D)), rng.standard_normal(N)
        gps = [_lib.DeviceGP(c, "matern52", N, D) for c in ctxs]
        multi.set_data(gps, X, y)
        multi.fit(gps, np.concatenate([[0.0], np.zeros(D), [-3.0]]), float(y.mean()))
        Xc = rng.random((M, D))
        shards = _lib.CandidateShards.split(ctxs, Xc)
        vals, mx, am, _, _ = multi.acq(gps, "ei", 0.0, float(y.min()), shards, True)
        one = gps[0].acq("ei", 0.0, float(y.min()), Xc, True)
        assert np.array_equal(vals, one[0]) and am == one[2] and mx == one[1], cycle
        shards.close()
        for g in gps:
            g.close()
        multi.close()
        for c in ctxs:
            c.close()
        if cycle == 7:
            n_tasks = tasks()
    assert _lib.live_contexts() == before
    assert tasks() <= n_tasks                                # worker threads end with their handle

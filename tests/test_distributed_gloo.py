"""The N > 1 exchange steps of the sharded hot path on CPU, world_size 2 (and 3, uneven shards): the library's own collective entry points
(robo_amd/csrc/comm.hip) through the interpreter build, with tests/hipemu/fake_rccl.cpp standing in for librccl.so
(ROBO_RCCL_LIB) and torch.distributed (gloo) doing what it does in a real job -- the rendezvous of the 128-byte id."""
import os
import socket

import numpy as np
import pytest

torch = pytest.importorskip("torch")
import torch.distributed as dist  # noqa: E402
import torch.multiprocessing as mp  # noqa: E402


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _emu_setup():
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(here, "hipemu"))
    import build_emu
    from robo_amd import _lib
    os.environ["ROBO_RCCL_LIB"] = build_emu.build_fake_rccl()
    _lib.use_library(build_emu.build())
    return _lib


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    _lib = _emu_setup()
    from robo_amd import sharding
    c, r, w = sharding.dist_info()               # the library communicator, id broadcast through the process group
    assert isinstance(c, _lib.Comm) and (r, w) == (rank, world)
    # candidate shard: global acquisition vector with a tie across the shard boundary and a NaN case
    M = 1000
    y = np.random.RandomState(5).randint(0, 50, size=M).astype(float)
    y[[123, 777]] = 99.0                        # tie: the lower global index must win
    b, e = sharding.shard_range(M, rank, world)
    j = int(np.argmax(y[b:e]))
    v, i = sharding.allgather_argmax(y[b + j], b + j)
    assert (v, i) == (99.0, 123), (v, i)
    y[900] = np.nan
    j = int(np.argmax(y[b:e]))
    v, i = sharding.allgather_argmax(y[b + j], b + j)
    assert i == 900 and np.isnan(v)
    # sample shard: rank-ordered sum of partial acquisition sums, identical on every rank
    S = 7
    acq = np.random.RandomState(6).rand(S, 64)
    sb, se = sharding.shard_range(S, rank, world)
    total = sharding.allgather_ordered_sum(acq[sb:se].sum(axis=0))
    np.testing.assert_allclose(total, acq.sum(axis=0), rtol=1e-14)
    np.save(os.path.join(out_dir, "total_%d.npy" % rank), total)
    # the fused entry points on device handles: candidate shard and sample shard against the single-rank calls
    from oracle import gp_oracle as O
    rs = np.random.RandomState(3)
    N, D, Mc = 150, 3, 701
    X = rs.rand(N, D)
    yy = np.sin(3 * X.sum(axis=1))
    Xc = rs.rand(Mc, D)
    thetas = np.array([[0.1, np.log(0.5), np.log(0.7), np.log(0.9), np.log(1e-2)]]) + 0.2 * rs.randn(5, 5)
    ctx = c.ctx
    gps = [_lib.DeviceGP(ctx, "matern52", N, D) for _ in range(5)]
    gps[0].set_data(X, yy)
    _, st = _lib.fit_batch(gps, thetas, float(yy.mean()))
    assert np.all(st == _lib.OK)
    eta = float(yy.min())
    full = _lib.Candidates(ctx, Xc)
    v_ref, mx_ref, am_ref, fl_ref = gps[0].acq("ei", 0.0, eta, full)
    b, e = sharding.shard_range(Mc, rank, world)
    mine = _lib.Candidates(ctx, Xc[b:e])
    v, mx, am, owner, fl = c.acq_sharded(gps[0], "ei", 0.0, eta, mine, b, want_values=True)
    np.testing.assert_array_equal(v, v_ref[b:e])
    owner_ref = [r for r in range(world) if sharding.shard_range(Mc, r, world)[0] <= am_ref < sharding.shard_range(Mc, r, world)[1]][0]
    assert (mx, am, fl) == (mx_ref, am_ref, fl_ref) and owner == owner_ref
    vm_ref, mxm_ref, amm_ref, _ = _lib.acq_marginal(gps, "log_ei", 0.0, np.full(5, eta), full)
    sb, se = sharding.shard_range(5, rank, world)
    vm, mxm, amm, _ = c.acq_marginal_sharded(gps[sb:se], 5, "log_ei", 0.0, np.full(se - sb, eta), full)
    np.testing.assert_allclose(vm, vm_ref, rtol=1e-13, atol=1e-15)
    assert amm == amm_ref
    np.save(os.path.join(out_dir, "marg_%d.npy" % rank), vm)
    # a rank without samples (S < world) takes part with an empty partial sum
    vm1, _, amm1, _ = c.acq_marginal_sharded(gps[:1] if rank == 0 else [], 1, "log_ei", 0.0, [eta], full)
    v1, _, am1, _ = gps[0].acq("log_ei", 0.0, eta, full)
    np.testing.assert_array_equal(vm1, v1)
    assert amm1 == am1
    # a rank whose local half fails still joins the exchange, and EVERY rank reports the failure (ranks that
    # disagree about the outcome of a collective call hang in the next one)
    unfitted = _lib.DeviceGP(ctx, "matern52", N, D)
    for bad in range(world):
        with pytest.raises(Exception, match="trained first"):
            c.acq_sharded(unfitted if rank == bad else gps[0], "ei", 0.0, eta, mine, b)
        with pytest.raises(Exception, match="trained first"):
            c.acq_marginal_sharded([unfitted] if rank == bad else gps[:1], world, "log_ei", 0.0, [eta], full)
    # ... and the communicator is still usable afterwards
    v2, mx2, am2, _, _ = c.acq_sharded(gps[0], "ei", 0.0, eta, mine, b, want_values=True)
    assert (mx2, am2) == (mx_ref, am_ref)
    for h in gps + [full, mine, unfitted]:
        h.close()
    dist.barrier()
    sharding.close_comm()
    dist.destroy_process_group()


def test_sharded_exchange_world2(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for name in ("total", "marg"):
        a = np.load(tmp_path / ("%s_0.npy" % name))
        b = np.load(tmp_path / ("%s_1.npy" % name))
        np.testing.assert_array_equal(a, b)        # bit-identical on both ranks


def test_sharded_exchange_world3(tmp_path):
    """uneven shards (701 candidates 234/234/233, 5 samples 2/2/1, 1 sample on rank 0 only): same checks, three ranks"""
    port = _free_port()
    mp.spawn(_worker, args=(3, port, str(tmp_path)), nprocs=3, join=True)
    for name in ("total", "marg"):
        a = np.load(tmp_path / ("%s_0.npy" % name))
        for r in (1, 2):
            np.testing.assert_array_equal(a, np.load(tmp_path / ("%s_%d.npy" % (name, r))))


def _class_worker(rank, world, port, out_dir):
    """the product classes under torch.distributed (gloo), device code through the interpreter build"""
    import sys
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    _lib = _emu_setup()
    from robo_amd import sharding
    from robo_amd import acquisition_functions as A
    from robo_amd.kernels import Matern52Kernel
    from robo_amd.maximizers import DeviceRandomSampling, RandomSampling
    from robo_amd.models import GaussianProcess, GaussianProcessMCMC
    from robo_amd.priors import DefaultPrior

    D = 3
    lo, hi = np.array([-1.0, 0.0, 2.0]), np.array([1.0, 5.0, 3.0])
    rs = np.random.RandomState(11)
    X = lo + (hi - lo) * rs.rand(30, D)
    y = np.sinc((X - lo) / (hi - lo) * 10 - 5).sum(axis=1)
    Xc = lo + (hi - lo) * np.random.RandomState(12).rand(301, D)      # 301: ragged shards (151 / 150)

    def mcmc(shard):
        kernel = 2 * Matern52Kernel(np.ones(D), ndim=D)
        m = GaussianProcessMCMC(kernel, prior=DefaultPrior(len(kernel) + 1, rng=np.random.RandomState(5)), n_hypers=10,
                                chain_length=3, burnin_steps=3, lower=lo, upper=hi, rng=np.random.RandomState(6))
        m.sample_shard = shard
        m.train(X, y, do_optimize=True)
        return m

    full, part = mcmc(False), mcmc(True)
    np.testing.assert_array_equal(np.array(full.hypers), np.array(part.hypers))     # replicated chain
    b, e = sharding.shard_range(10, rank, world)
    assert [m.is_trained for m in part.models] == [b <= i < e for i in range(10)]
    assert all(m.gp is None for i, m in enumerate(part.models) if not b <= i < e)   # no device memory off-shard
    # sample shard: marginal acquisition and mixture posterior == the single-rank result
    for cls in (A.LogEI, A.EI, A.LCB):
        ref = A.MarginalizationGPMCMC(cls(full))
        sh = A.MarginalizationGPMCMC(cls(part))
        sh.sample_shard = True
        want, got = ref.compute(Xc), sh.compute(Xc)
        np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-14)
        assert sh.argmax(Xc) == ref.argmax(Xc) == int(np.argmax(want))
    m_ref, v_ref = full.predict(Xc)
    m_sh, v_sh = part.predict(Xc)
    np.testing.assert_allclose(m_sh, m_ref, rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(v_sh, v_ref, rtol=1e-9, atol=1e-12)
    # candidate shard: RandomSampling under a process group == plain np.argmax over the same candidates
    gp = GaussianProcess(2 * Matern52Kernel(np.ones(D), ndim=D), noise=1e-3, lower=lo, upper=hi,
                         rng=np.random.RandomState(7))
    gp.train(X, y, do_optimize=False)
    acq = A.EI(gp)
    np.random.seed(21)
    x_sharded = RandomSampling(acq, lo, hi, n_samples=203, rng=np.random.RandomState(8), shard=True).maximize()
    np.random.seed(21)
    cands = RandomSampling(acq, lo, hi, n_samples=203, rng=np.random.RandomState(8)).candidates()
    np.testing.assert_array_equal(x_sharded, cands[int(np.argmax(acq.compute(cands)))])
    assert sharding.sharded_argmax(acq, Xc) == int(np.argmax(acq.compute(Xc)))
    # fewer candidates than ranks: the rank with an empty slice issues the same collective as the others
    assert sharding.sharded_argmax(acq, Xc[:1]) == 0
    assert sharding.sharded_argmax(lambda Z: -np.abs(Z).sum(axis=1), Xc[:1]) == 0     # generic (host-vector) form
    # device-generated candidates: per-rank Philox shards, the winner's point travels to every rank
    x_dev = DeviceRandomSampling(acq, lo, hi, n_samples=1001, rng=np.random.RandomState(9), shard=True).maximize()
    assert np.all(x_dev >= lo) and np.all(x_dev <= hi)
    np.save(os.path.join(out_dir, "xdev_%d.npy" % rank), x_dev)
    # ... and it is the best point of the union of the shards, re-evaluated here on one rank
    best = (-np.inf, None)
    seed = int(np.random.RandomState(9).randint(0, 2 ** 31 - 1))
    for r in range(world):
        rb, re = sharding.shard_range(1001, r, world)
        c = _lib.Candidates(gp.gp.ctx, m=re - rb, seed=seed + 7919 * r, n_uniform=min(max(700 - rb, 0), re - rb),
                            loc=(gp.get_incumbent()[0] - lo) / (hi - lo), scale=0.1 / (hi - lo))
        pts = lo + (hi - lo) * c.points()
        vals = acq.compute(pts)
        j = int(np.argmax(vals))
        if vals[j] > best[0]:
            best = (vals[j], pts[j])
        c.close()
    np.testing.assert_allclose(x_dev, best[1], rtol=1e-13)
    # Sobol candidates: per-rank slices of ONE sequence -> exactly the single-process winner
    from robo_amd.maximizers import DeviceSobolSampling
    from scipy.stats import qmc
    x_sob = DeviceSobolSampling(acq, lo, hi, n_samples=513, seed=4, shard=True).maximize()
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        pts = lo + (hi - lo) * qmc.Sobol(d=D, scramble=True, seed=4).random(513)
    np.testing.assert_array_equal(x_sob, pts[int(np.argmax(acq.compute(pts)))])
    # without the explicit opt-in a process group changes nothing: every rank maximises over all candidates itself
    np.random.seed(21)
    x_plain = RandomSampling(acq, lo, hi, n_samples=203, rng=np.random.RandomState(8)).maximize()
    np.testing.assert_array_equal(x_plain, x_sharded)
    # ranks that drifted apart (different seeds) are caught before the exchange, on every rank
    import pytest
    with pytest.raises(RuntimeError, match="differ across ranks"):
        np.random.seed(100 + rank)
        RandomSampling(acq, lo, hi, n_samples=203, rng=np.random.RandomState(8), shard=True).maximize()
    with pytest.raises(RuntimeError, match="differ across ranks"):
        DeviceSobolSampling(acq, lo, hi, n_samples=513, seed=4 + rank, shard=True).maximize()
    # large offsets between the per-sample means: the two-pass variance of the sharded mixture stays accurate
    big = mcmc(True)
    for i, mdl in enumerate(big.models):
        if mdl.is_trained:
            mdl.gp.set_output_transform(1.0e8 + i, 1.0)       # mu_s -> mu_s + 1e8 + s
    for i, mdl in enumerate(full.models):
        mdl.gp.set_output_transform(1.0e8 + i, 1.0)
    m_ref, v_ref = full.predict(Xc)
    m_sh, v_sh = big.predict(Xc)
    np.testing.assert_allclose(m_sh, m_ref, rtol=1e-13)
    np.testing.assert_allclose(v_sh, v_ref, rtol=1e-9)
    # candidate shard of the information gain per unit cost (BASELINE config 4: Fabolas objective + cost models): the
    # global argmax of robo_ig_eval_per_cost_cand_sharded == the single-rank np.argmax of the same candidates
    from robo_amd.kernels import FabolasKernel
    from robo_amd.models.fabolas_gp import FabolasGP
    rs = np.random.RandomState(31)
    Xf = np.concatenate([lo[:2] + (hi[:2] - lo[:2]) * rs.rand(40, 2), rs.rand(40, 1) * 0.9 + 0.1], axis=1)
    yf = np.sin(Xf[:, 0]) + 0.2 * (1 - Xf[:, 2]) ** 2
    cf = np.log(0.2 + 3.0 * Xf[:, 2])
    obj = FabolasGP(FabolasKernel(3, metric=0.5), basis_function=lambda s_: (1 - s_) ** 2, noise=1e-3, lower=lo[:2],
                    upper=hi[:2], rng=np.random.RandomState(32))
    cost = FabolasGP(FabolasKernel(3, metric=0.5), basis_function=lambda s_: s_, noise=1e-3, lower=lo[:2], upper=hi[:2],
                     rng=np.random.RandomState(33))
    obj.train(Xf, yf, do_optimize=False)
    cost.train(Xf, cf, do_optimize=False)
    is_env = np.array([0, 0, 1])
    np.random.seed(34)
    igc = A.InformationGainPerUnitCost(obj, cost, np.append(lo[:2], 0), np.append(hi[:2], 1), is_env,
                                       sampling_acquisition=A.EI, n_representer=10, Np=50,
                                       rng=np.random.RandomState(35))
    igc.shard = True                    # rank 0's representer points on every rank (the sampler's stream is OS-seeded)
    igc.update(obj, cost, overhead=0.3)
    np.testing.assert_array_equal(sharding.allgather_rows(np.asarray(igc.zb).ravel())[0], np.asarray(igc.zb).ravel())
    Xq = np.concatenate([lo[:2] + (hi[:2] - lo[:2]) * np.random.RandomState(36).rand(77, 2),
                         np.random.RandomState(37).rand(77, 1)], axis=1)
    want = igc.compute(Xq)
    assert igc._native_cost() and igc.argmax(Xq) == int(np.argmax(want))
    assert sharding.sharded_argmax(igc, Xq) == int(np.argmax(want))
    assert sharding.sharded_argmax(igc, Xq[:1]) == 0                 # one candidate, two ranks: an empty shard joins
    # the solver loop in this form (every rank runs it in lock step): the objective is evaluated on rank 0 ONLY -- a
    # rank-dependent (non-deterministic) objective would otherwise give the ranks different data -- and an entropy-search
    # acquisition whose `shard` flag was forgotten is switched on by the loop (rank 0's representer points everywhere)
    from robo_amd.solver import BayesianOptimization
    from robo_amd.initial_design import init_latin_hypercube_sampling
    calls = []

    def objective(x):
        calls.append(1)
        return float(np.sum((x - 1.0) ** 2)) + 1000.0 * rank          # what rank 1 would return must never be seen

    kernel2 = 2 * Matern52Kernel(np.ones(2), ndim=2)
    gp2 = GaussianProcess(kernel2, lower=lo[:2], upper=hi[:2], rng=np.random.RandomState(40))
    ig2 = A.InformationGain(gp2, lo[:2], hi[:2], Nb=8, Np=20, sampling_acquisition=A.EI, rng=np.random.RandomState(41))
    assert ig2.shard is False
    np.random.seed(42)
    bo = BayesianOptimization(objective, lo[:2], hi[:2], ig2, gp2,
                              RandomSampling(ig2, lo[:2], hi[:2], n_samples=60, rng=np.random.RandomState(43), shard=True),
                              initial_design=init_latin_hypercube_sampling, initial_points=3, rng=np.random.RandomState(44))
    bo.run(5)
    assert len(calls) == (5 if rank == 0 else 0), (rank, len(calls))
    assert ig2.shard is True
    rows = sharding.allgather_rows(np.concatenate([bo.X.ravel(), bo.y.ravel(), np.asarray(ig2.zb).ravel()]))
    np.testing.assert_array_equal(rows[0], rows[-1])
    assert np.all(bo.y < 500.0)
    dist.barrier()
    sharding.close_comm()
    dist.destroy_process_group()


def test_product_classes_sharded_world2(tmp_path):
    """GaussianProcessMCMC / MarginalizationGPMCMC sample shard and RandomSampling / DeviceRandomSampling candidate
    shard under torch.distributed (world_size 2, gloo): equal to the single-rank results"""
    port = _free_port()
    mp.spawn(_class_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    np.testing.assert_array_equal(np.load(tmp_path / "xdev_0.npy"), np.load(tmp_path / "xdev_1.npy"))

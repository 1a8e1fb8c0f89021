"""CPU-side checks of the HIP sources' LOGIC through the g++ lockstep interpreter
(tests/hipemu): index arithmetic, tiling, barrier structure, fragment maps as documented,
C-ABI orchestration and the Python host shims.  NOT a parity claim for the product -- the
parity tests proper are tests/test_gpu_parity.py (-m gpu, real MI355X).
"""
import os
import sys

import numpy as np
import pytest

import parity_checks as P
from robo_amd import _lib

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def emu_ctx():
    sys.path.insert(0, os.path.join(HERE, "hipemu"))
    import build_emu
    path = build_emu.build()
    _lib.use_library(path)
    ctx = _lib.Context(0)
    assert "hipemu" in ctx.name
    yield ctx
    ctx.close()
    _lib.use_library(None)


def test_mfma_fragment_map_as_documented(emu_ctx):
    assert emu_ctx.selftest_mfma_layout() < 1e-12


@pytest.mark.parametrize("name", ["small_matern", "ragged_rbf_nout", "one_block_edge", "two_block"])
def test_golden_cases(emu_ctx, name):
    P.check_case(emu_ctx, name)


def test_mcmc_marginal(emu_ctx):
    P.check_mcmc_marginal(emu_ctx)


def test_mcmc_draws_equal_numpy_legacy_stream(emu_ctx):
    """robo_mcmc_draws == the RandomState calls of emcee 2's loop (rand, randint, rand per half-step): numbers AND the
    state the stream is left in, across the MT19937 block boundary, with a cached Gaussian kept, for odd sizes"""
    for seed, n_steps, half, burn in ((0, 7, 5, 0), (3, 40, 26, 11), (9, 3, 1, 617), (11, 25, 64, 1000), (2, 0, 4, 3)):
        a, b = np.random.RandomState(seed), np.random.RandomState(seed)
        for r in (a, b):
            r.rand(burn)
            r.randn(1)                      # leaves a cached Gaussian in the state: must survive
        uz = np.empty((n_steps, 2, half)); ua = np.empty((n_steps, 2, half)); pa = np.empty((n_steps, 2, half), dtype=np.int64)
        for it in range(n_steps):
            for h in range(2):
                uz[it, h] = a.rand(half)
                pa[it, h] = a.randint(half, size=(half,))
                ua[it, h] = a.rand(half)
        uz2, pa2, ua2 = _lib.mcmc_draws(b, n_steps, half)
        np.testing.assert_array_equal(uz2, uz)
        np.testing.assert_array_equal(pa2, pa)
        np.testing.assert_array_equal(ua2, ua)
        sa, sb = a.get_state(), b.get_state()
        assert sa[0] == sb[0] and np.array_equal(sa[1], sb[1]) and sa[2:] == sb[2:]
        assert a.rand() == b.rand() and a.randn() == b.randn()


def test_device_resident_chain(emu_ctx):
    # (the small-N chains of the reference's own front-end run are replayed in tests/test_ref_parity.py; here the shapes:
    # two panels, the three kernel kinds, both priors)
    P.check_device_chain(emu_ctx, cases=(("matern52", 150, 3, 10, 4), ("rbf", 40, 2, 8, 5),
                                               ("fabolas", 50, 3, 12, 4), ("fabolas", 60, 3, 12, 3, "env"),
                                               ("fabolas", 150, 4, 14, 2, "env")))


def test_elementwise_and_degenerate_branches(emu_ctx):
    P.check_elementwise(emu_ctx)


def test_argmax_semantics(emu_ctx):
    P.check_argmax_semantics(emu_ctx)


def test_error_protocol(emu_ctx):
    P.check_errors(emu_ctx)


def test_edge_sizes(emu_ctx):
    P.check_edge_sizes(emu_ctx)


def test_uniform_generator(emu_ctx):
    P.check_uniform_generator(emu_ctx)


def test_fit_batch_keeps_factors(emu_ctx):
    P.check_fit_batch(emu_ctx)
    P.check_fit_batch(emu_ctx, sizes=((90, 4),), kind="fabolas")


def test_batched_likelihoods(emu_ctx):
    P.check_batched_likelihoods(emu_ctx)


def test_batched_split_streams(emu_ctx):
    """sub-batches on side streams with staggered group boundaries: same bits (5 panels at N = 520)"""
    P.check_batched_split(emu_ctx)


def test_batched_fit_multiple_of_128(emu_ctx):
    """the thin tiles of the augmented row's block row, in the 32-row and (threshold lowered) the 128-row update kernel"""
    P.check_batched_multiple_of_128(emu_ctx, sizes=((256, 3),))
    P.check_batched_multiple_of_128(emu_ctx, sizes=((384, 2),), S=4, tm4_min=1)


def test_context_lifetime(emu_ctx):
    """a context closed BEFORE the handles that live on it (what a garbage collector may do: Python's cyclic GC finalises
    an unreachable group in no particular order) stays usable until the last of them is destroyed, then goes"""
    n0 = _lib.live_contexts()
    ctx = _lib.Context(0)
    rs = np.random.RandomState(0)
    X, y, Xc = rs.rand(30, 2), rs.rand(30), rs.rand(9, 2)
    theta = np.array([0.0, -0.5, -0.5, np.log(1e-2)])
    gp = _lib.DeviceGP(ctx, "matern52", 30, 2)
    gp.set_data(X, y)
    cand = _lib.Candidates(ctx, Xc)
    want = (gp.fit(theta, 0.0), gp.predict(cand))
    assert _lib.live_contexts() == n0 + 1
    ctx.close()                                         # marked, not released: two handles live on it
    assert _lib.live_contexts() == n0 + 1
    assert gp.fit(theta, 0.0) == want[0]
    np.testing.assert_array_equal(gp.predict(cand)[0], want[1][0])
    gp.close()
    assert _lib.live_contexts() == n0 + 1
    cand.close()                                        # the last handle takes the context with it
    assert _lib.live_contexts() == n0
    # the ordinary order, and a Multi over two contexts closed before it
    a, b = _lib.Context(0), _lib.Context(0)
    g2 = _lib.DeviceGP(a, "matern52", 30, 2)
    g2.close()
    multi = _lib.Multi([a, b])
    a.close()
    b.close()
    assert _lib.live_contexts() == n0 + 2
    multi.close()
    assert _lib.live_contexts() == n0


def test_grad_loglik(emu_ctx):
    P.check_grad_loglik(emu_ctx)


def test_model_gradients(emu_ctx):
    P.check_model_gradients(emu_ctx)


def test_ill_conditioned(emu_ctx):
    P.check_ill_conditioned(emu_ctx)


def test_device_random_candidates(emu_ctx):
    P.check_device_random_candidates(emu_ctx)


def test_shape_sweep(emu_ctx):
    P.check_shape_sweep(emu_ctx, n_cases=6, seed=5, max_n=200)


def test_fabolas_kernel(emu_ctx):
    P.check_fabolas_kernel(emu_ctx)


def test_fp32_gram_mixed_precision(emu_ctx):
    P.check_fp32_gram(emu_ctx)


def test_chunked_workspace_equals_single_pass(emu_ctx, monkeypatch):
    """candidate batches larger than the solve workspace are processed in chunks"""
    P.check_chunked_workspace(emu_ctx, monkeypatch, N=150, D=3, M=700, ws_blocks=2)



def test_multi_panel_factorisation(emu_ctx):
    """N = 520 -> 5 panels (panel solves on 32-row tiles, trailing updates on the small-tile
    variant); fit only (the interpreter is slow).  The 64- and 128-row variants run on the GPU."""
    from oracle import gp_oracle as O
    rs = np.random.RandomState(12)
    N, D = 520, 3
    X = rs.rand(N, D)
    y = np.sin(4 * X.sum(axis=1))
    theta = np.array([0.2, np.log(0.4), np.log(0.6), np.log(0.8), np.log(1e-2)])
    ogp = O.OracleGP("matern52", theta, normalize_input=False)
    ogp.train(X, y)
    g = _lib.DeviceGP(emu_ctx, "matern52", N, D)
    g.set_data(X, y)
    ll = g.fit(theta, ogp.mean)
    np.testing.assert_allclose(ll, ogp.loglikelihood(theta), rtol=1e-10)
    np.testing.assert_allclose(g.factor(), ogp.L, rtol=0, atol=1e-11)
    g.close()


def test_persistent_tile_updates(emu_ctx, monkeypatch):
    """the 128-row trailing update with several tiles per workgroup (next tile's C prefetched while
    the current one is multiplied): forced at N = 520 through the launcher's test knobs; same bits
    as the one-tile-per-workgroup schedule"""
    from oracle import gp_oracle as O
    rs = np.random.RandomState(13)
    N, D = 520, 3
    X = rs.rand(N, D)
    y = np.cos(3 * X.sum(axis=1))
    theta = np.array([0.1, np.log(0.5), np.log(0.7), np.log(0.9), np.log(1e-2)])
    ogp = O.OracleGP("matern52", theta, normalize_input=False)
    ogp.train(X, y)
    g = _lib.DeviceGP(emu_ctx, "matern52", N, D)
    g.set_data(X, y)
    g.fit(theta, ogp.mean)
    L_ref = g.factor().copy()
    try:
        emu_ctx.set_tuning("potrf_tm4_min", 1)
        # step 0 has 10 tiles (9 for the tile workgroups).  cap 4: 3 workgroups x 3 rounds; cap 3: 2 workgroups, 4 rounds +
        # 1 tile -> as two 64-row halves; cap 5: 4 workgroups, 2 rounds + 1 tile -> as four 32-row quarters; cap 9: 8
        # workgroups, 1 round + 1 tile -> quarters; cap 64: one tile each.  With the tail split off: whole tiles only.
        for split in (1, 0):
            emu_ctx.set_tuning("potrf_tail_split", split)
            for cap in (4, 3, 5, 9, 64):
                emu_ctx.set_tuning("potrf_max_wg", cap)
                ll = g.fit(theta, ogp.mean)
                np.testing.assert_allclose(ll, ogp.loglikelihood(theta), rtol=1e-10)
                np.testing.assert_array_equal(g.factor(), L_ref)
    finally:
        emu_ctx.set_tuning("potrf_tm4_min", None)
        emu_ctx.set_tuning("potrf_max_wg", None)
        emu_ctx.set_tuning("potrf_tail_split", None)
    np.testing.assert_allclose(L_ref, ogp.L, rtol=0, atol=1e-11)
    g.close()


def test_predictive_gradients(emu_ctx):
    P.check_predictive_gradients(emu_ctx, cases=(("matern52", 70, 3, 9), ("fabolas", 60, 3, 7)))


def test_sobol_candidates(emu_ctx):
    P.check_sobol_candidates(emu_ctx, dims=(3, 17), m=300 + 900)


def test_candidate_reupload(emu_ctx):
    P.check_candidate_reupload(emu_ctx)


def test_phase_events(emu_ctx):
    P.check_phase_events(emu_ctx)


def test_small_and_large_candidate_tiles_agree(emu_ctx, monkeypatch):
    P.check_small_and_large_tiles_agree(emu_ctx, monkeypatch)


def test_host_array_handle_reuse(emu_ctx):
    P.check_host_array_handle_reuse(emu_ctx)



def test_winv_small_batch_path(emu_ctx):
    """the explicit-inverse posterior for small batches (winv.hip): index arithmetic, unit table, chunk order"""
    P.check_winv_path(emu_ctx, cases=(("matern52", 300, 5, 200), ("fabolas", 280, 4, 130)))


def test_winv_condition_guard_sweep(emu_ctx):
    """the cond_inf(L) guard of the explicit-inverse posterior (same arithmetic as on the MI355X, three block rows)"""
    P.check_winv_guard_sweep(emu_ctx, n=384, min_blocks=2, m=200,
                             sweep=((2, (1e-3, 1e-9)), (1, (1e-7, 1e-10, 1e-12))))


def test_george_kernel_api_slice(emu_ctx):
    """the slice of george's kernel API the reference and its tests use (robo_amd/kernels.py): a kernel WITHOUT an
    amplitude factor has no amplitude parameter (len = D; the library's log amp is pinned at 0), ``b * kernel`` has one
    (len = 1 + D); ``axes`` covering all columns; ``get_value`` for one and for two point sets"""
    from oracle import gp_oracle as O
    from robo_amd.kernels import ExpSquaredKernel, Matern52Kernel
    from robo_amd.models import GaussianProcess, GaussianProcessMCMC
    rs = np.random.RandomState(8)
    X, Z = rs.rand(14, 2), rs.rand(5, 2)
    y = np.sin(4 * X.sum(axis=1))
    bare, full = Matern52Kernel(np.array([0.3, 0.6]), ndim=2), 1.0 * Matern52Kernel(np.array([0.3, 0.6]), ndim=2)
    assert len(bare) == 2 and len(full) == 3 and bare.fixed_head() == (0.0,) and full.fixed_head() == ()
    np.testing.assert_array_equal(bare.get_parameter_vector(), np.log([0.3, 0.6]))
    np.testing.assert_array_equal(bare[:], np.log([0.3, 0.6]))
    bare.set_parameter_vector(np.log([0.4, 0.5]))
    assert len((2 * bare)) == 3 and (2 * bare).get_parameter_vector()[0] == np.log(2.0 / 2)
    np.testing.assert_array_equal((2 * bare).get_parameter_vector()[1:], np.log([0.4, 0.5]))
    bare.set_parameter_vector(np.log([0.3, 0.6]))
    assert len(Matern52Kernel(np.array([1]), axes=0, ndim=1)) == 1 and len(ExpSquaredKernel(0.5, ndim=2, axes=[0, 1])) == 2
    with pytest.raises(NotImplementedError):
        len(Matern52Kernel(np.array([0.01]), ndim=3, axes=1))     # a single factor of the Fabolas product: no kernel on its own
    # values: the oracle's kernel with amplitude 1
    theta_full = np.concatenate([[0.0], np.log([0.3, 0.6])])
    np.testing.assert_allclose(bare.get_value(X), O.kernel_matrix("matern52", theta_full, X, X), rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(bare.get_value(Z, X), O.kernel_matrix("matern52", theta_full, Z, X), rtol=1e-12, atol=1e-14)
    # the two models with both kernels: same numbers, one parameter fewer
    ga = GaussianProcess(bare, noise=1e-3, lower=np.zeros(2), upper=np.ones(2))
    gb = GaussianProcess(1.0 * Matern52Kernel(np.array([0.3, 0.6]), ndim=2) * 2.0, noise=1e-3, lower=np.zeros(2), upper=np.ones(2))
    ga.train(X, y, do_optimize=False)
    gb.train(X, y, do_optimize=False)
    assert ga.hypers.shape == (3,) and gb.hypers.shape == (4,)
    t3 = np.array([np.log(0.2), np.log(0.7), np.log(1e-2)])
    t4 = np.concatenate([[0.0], t3])
    assert ga.nll(t3) == gb.nll(t4)
    np.testing.assert_array_equal(ga.grad_nll(t3), gb.grad_nll(t4)[1:])
    gb.kernel.set_parameter_vector(np.concatenate([[0.0], np.log([0.3, 0.6])]))
    gb.train(X, y, do_optimize=False)
    ga.train(X, y, do_optimize=False)       # (nll() leaves the handle fitted at ITS theta, as the reference's does)
    np.testing.assert_array_equal(ga.predict(Z)[0], gb.predict(Z)[0])
    assert ga.optimize().shape == (3,)
    # test/test_models/test_gaussian_process_mcmc.py:16-23: six walkers are enough for D + 1 = 3 hyper-parameters
    mc = GaussianProcessMCMC(Matern52Kernel(np.ones(2), ndim=2), n_hypers=6, burnin_steps=3, chain_length=4,
                             rng=np.random.RandomState(1))
    mc.train(X, y, do_optimize=True)
    assert np.asarray(mc.hypers).shape == (6, 3) and len(mc.models) == 6
    assert np.isfinite(mc.loglikelihood(np.array([0.2, 0.2, 0.001])))
    m, v = mc.predict(Z)
    assert m.shape == (5,) and np.all(v > 0)


def test_grid_search_is_one_batched_call(emu_ctx):
    """robo/maximizers/grid_search.py: the grid scored in ONE call picks the point the reference's point-by-point loop picks"""
    from robo_amd.acquisition_functions import EI, LCB
    from robo_amd.kernels import Matern52Kernel
    from robo_amd.maximizers import GridSearch
    from robo_amd.models import GaussianProcess
    rs = np.random.RandomState(3)
    lo, hi = np.array([0.0]), np.array([6.0])
    X = rs.rand(8, 1) * 6
    y = np.sin(3 * X[:, 0]) * 4 * (X[:, 0] - 1) * (X[:, 0] + 2)
    model = GaussianProcess(2 * Matern52Kernel(np.ones(1), ndim=1), noise=1e-3, lower=lo, upper=hi)
    model.train(X, y, do_optimize=False)
    for acq in (EI(model), LCB(model)):
        acq.update(model)
        g = GridSearch(acq, lo, hi, resolution=200)
        grid = np.linspace(0.0, 6.0, 200)
        one_by_one = np.array([float(np.asarray(acq(np.array([[x]]))).reshape(-1)[0]) for x in grid])
        x = g.maximize()
        assert x.shape == (1,) and x[0] == grid[int(one_by_one.argmax())]
    with pytest.raises(RuntimeError):
        GridSearch(acq, np.zeros(2), np.ones(2))

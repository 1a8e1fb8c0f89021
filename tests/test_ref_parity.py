"""robo_amd's classes against fixtures made by the REFERENCE'S OWN GP-side classes (tests/ref_checks.py).

-m gpu: through librobo_hip.so on the MI355X (the parity claim).  The `emulated` variants push the small
cases through tests/hipemu on the CPU: they check host logic (normalisation, sampler draw order, quirks),
nothing about the hardware.
"""
import os
import sys

import numpy as np
import pytest

import ref_checks as R
from robo_amd import _lib

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def emu():
    sys.path.insert(0, os.path.join(HERE, "hipemu"))
    import build_emu
    _lib.use_library(build_emu.build())
    yield
    _lib.use_library(None)


@pytest.fixture(scope="module")
def gpu():
    _lib.use_library(None)
    if _lib.device_count() < 1:
        pytest.skip("no HIP device")
    yield


# ---- CPU (interpreter): host logic -----------------------------------------------------------------
@pytest.mark.parametrize("name", ["ref_gp_matern", "ref_gp_rbf_nout"])
def test_gp_class_emulated(emu, name):
    R.check_ref_gp(name)


def test_gp_retry_emulated(emu):
    R.check_ref_gp_retry()


def test_mcmc_chain_and_marginal_emulated(emu):
    R.check_ref_mcmc()


def test_fabolas_emulated(emu):
    R.check_ref_fabolas()


def test_infogain_emulated(emu):
    R.check_ref_infogain()


def test_infogain_per_unit_cost_emulated(emu):
    R.check_ref_infogain_cost()


def test_branin_replay_emulated(emu):
    R.check_ref_branin_replay()


def test_single_point_maximizers_replay_emulated(emu):
    """robo.fmin.bayesian_optimization(maximizer="scipy" / "differential_evolution"): the reference's own two runs replayed"""
    checked, same = R.check_ref_single_point_replay(max_iters=2)       # (the interpreter is slow; all 16 on the MI355X)
    assert checked == 4 and same >= 3, (checked, same)


def test_gp_mcmc_front_end_replay_emulated(emu):
    """robo.fmin.bayesian_optimization(model_type="gp_mcmc", acquisition_func="log_ei"): the reference's own run replayed --
    the candidate chosen at all 8 model-based iterations, and robo_amd's own chains ending on the reference's walkers"""
    checked, gap = R.check_ref_branin_gpmcmc_replay(chain=False)
    assert checked == 8 and gap > 1e-7, (checked, gap)


def test_gp_mcmc_front_end_free_run_emulated(emu):
    """robo_amd.fmin.bayesian_optimization(model_type="gp_mcmc") left to itself with the reference's seeds returns the
    reference's run: every evaluated point, bit for bit (7 of the 11 iterations here, all of them on the MI355X)"""
    assert R.check_ref_branin_gpmcmc_free_run(num_iterations=7) == 7


@pytest.mark.parametrize("acq", ["ei", "pi", "lcb"])
def test_gp_mcmc_front_end_other_acquisitions_free_run_emulated(emu, acq):
    """the same front end with EI / PI / LCB under MarginalizationGPMCMC: the reference's runs (fixture
    ref_branin_gpmcmc_acq), first 5 of 8 iterations here, bit for bit"""
    assert R.check_ref_branin_gpmcmc_free_run(num_iterations=5, acquisition_func=acq) == 5


def test_entropy_search_replay_emulated(emu):
    assert R.check_ref_entropy_search_replay() == 6


def test_entropy_search_default_model_replay_emulated(emu):
    """robo.fmin.entropy_search with its DEFAULT model (gp_mcmc): MarginalizationGPMCMC over 10 InformationGain estimators,
    the reference's choice bit for bit (first of the 3 model-based iterations here; all of them on the MI355X)"""
    assert R.check_ref_entropy_search_gpmcmc_replay(max_iters=1) == 1


def test_fabolas_replay_emulated(emu):
    assert R.check_ref_fabolas_replay(n_iter=1) == 1         # host logic; all three iterations run on the MI355X


# ---- MI355X ---------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("name", ["ref_gp_matern", "ref_gp_rbf_nout", "ref_gp_headline_shape"])
def test_gp_class(gpu, name):
    R.check_ref_gp(name)


@pytest.mark.gpu
def test_gp_retry_and_optimize(gpu):
    R.check_ref_gp_retry()
    mine, ref = R.check_ref_gp_optimize()
    print("optimised hypers", mine, "reference", ref)


@pytest.mark.gpu
def test_mcmc_chain_and_marginal(gpu):
    R.check_ref_mcmc()


@pytest.mark.gpu
def test_fabolas(gpu):
    R.check_ref_fabolas()


@pytest.mark.gpu
def test_infogain_every_candidate(gpu):
    R.check_ref_infogain()


@pytest.mark.gpu
def test_infogain_per_unit_cost(gpu):
    R.check_ref_infogain_cost()


@pytest.mark.gpu
def test_infogain_config4_shape(gpu):
    R.check_ref_infogain_config4()


@pytest.mark.gpu
def test_branin_trajectory_replay(gpu):
    assert R.check_ref_branin_replay() == 27


@pytest.mark.gpu
def test_branin_free_run(gpu):
    same, f_mine, f_ref = R.check_ref_branin_free_run()
    print("free run: identical choices for the first %d iterations; f_opt %.6f (reference %.6f)" % (same, f_mine, f_ref))
    # measured on the MI355X (r02a): the first 15 choices are identical although every iteration runs its own
    # finite-difference L-BFGS-B; after the first diverging optimiser run the two are different random searches
    assert same >= 8
    assert f_mine - 0.397887 <= 1.0


@pytest.mark.gpu
def test_single_point_maximizers_trajectory_replay(gpu):
    checked, same = R.check_ref_single_point_replay()
    print("single-point maximisers: %d iterations replayed, %d on the reference's point to 1e-3 of the box" % (checked, same))
    assert checked == 16 and same >= 12, (checked, same)


@pytest.mark.gpu
def test_entropy_search_trajectory_replay(gpu):
    """robo.fmin.entropy_search's own run (model="gp"): same choice at all 6 model-based iterations"""
    assert R.check_ref_entropy_search_replay() == 6


@pytest.mark.gpu
def test_fabolas_trajectory_replay(gpu):
    """robo.fmin.fabolas's own run: projected incumbents and the choice at all 3 model-based iterations"""
    assert R.check_ref_fabolas_replay() == 3


# (added after the round-5 GPU budget was spent: verified through the interpreter only, hence LAST in the last file --
# under `-x` a surprise here cannot hide another test)
@pytest.mark.gpu
def test_gp_mcmc_front_end_trajectory_replay(gpu):
    """the reference's own gp_mcmc + LogEI run: same choice at all 8 model-based iterations (marginal LogEI over the 10
    walkers, one batched fit per iteration), and robo_amd's device-resident chains end on the reference's walkers"""
    checked, gap = R.check_ref_branin_gpmcmc_replay()
    print("gp_mcmc front end: %d iterations replayed, smallest best-vs-second gap %.2e" % (checked, gap))
    assert checked == 8 and gap > 1e-7, (checked, gap)
    # and left to itself with the reference's seeds: the reference's whole result, bit for bit
    assert R.check_ref_branin_gpmcmc_free_run() == 11


@pytest.mark.gpu
@pytest.mark.parametrize("acq", ["ei", "pi", "lcb"])
def test_gp_mcmc_front_end_other_acquisitions_free_run(gpu, acq):
    """robo_amd.fmin.bayesian_optimization(model_type="gp_mcmc", acquisition_func=ei|pi|lcb) with the reference's seeds:
    the reference's whole result, bit for bit"""
    assert R.check_ref_branin_gpmcmc_free_run(acquisition_func=acq) == 8


@pytest.mark.gpu
def test_entropy_search_default_model_trajectory_replay(gpu):
    """robo.fmin.entropy_search's DEFAULT configuration (model="gp_mcmc"): the reference's own run, same choice at all 3
    model-based iterations (best-vs-second gaps of the marginal information gain 12-45 %)"""
    assert R.check_ref_entropy_search_gpmcmc_replay() == 3

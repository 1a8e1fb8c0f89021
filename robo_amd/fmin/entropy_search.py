"""``entropy_search`` front end with the signature of robo/fmin/entropy_search.py:20-131:
GP / GP-MCMC model + InformationGain (representer proposal: EI) + RandomSampling, run by the
BO loop.  The information gain of the whole candidate batch is one device call."""
import numpy as np

from robo_amd.acquisition_functions import EI, InformationGain, MarginalizationGPMCMC
from robo_amd.initial_design import init_latin_hypercube_sampling
from robo_amd.kernels import Matern52Kernel
from robo_amd.maximizers import DifferentialEvolution, RandomSampling, SciPyOptimizer
from robo_amd.models import GaussianProcess, GaussianProcessMCMC
from robo_amd.priors import DefaultPrior
from robo_amd.solver import BayesianOptimization


def build_entropy_search(lower, upper, maximizer="random", model="gp_mcmc", rng=None, n_candidates=500,
                         chain_length=200, burnin_steps=100, n_representer=50, n_outcomes=400, devices=None):
    """the objects robo/fmin/entropy_search.py:69-121 wires together -> (model, acquisition function, maximiser)"""
    n_dims = lower.shape[0]
    kernel = 2 * Matern52Kernel(np.ones([n_dims]), ndim=n_dims)
    prior = DefaultPrior(len(kernel) + 1)
    n_hypers = 3 * len(kernel)
    if n_hypers % 2 == 1:
        n_hypers += 1
    if model == "gp":
        gp = GaussianProcess(kernel, prior=prior, rng=rng, normalize_output=False, normalize_input=True, lower=lower,
                             upper=upper, devices=devices)
    elif model == "gp_mcmc":
        gp = GaussianProcessMCMC(kernel, prior=prior, n_hypers=n_hypers, chain_length=chain_length,
                                 burnin_steps=burnin_steps, normalize_input=True, normalize_output=False, rng=rng,
                                 lower=lower, upper=upper, devices=devices)
    else:
        raise ValueError("%s is not a valid model!" % model)
    a = InformationGain(gp, lower=lower, upper=upper, sampling_acquisition=EI, Nb=n_representer, Np=n_outcomes, rng=rng)
    acquisition_func = MarginalizationGPMCMC(a) if model == "gp_mcmc" else a
    if maximizer == "random":
        max_func = RandomSampling(acquisition_func, lower, upper, n_samples=n_candidates, rng=rng)
    elif maximizer == "scipy":
        max_func = SciPyOptimizer(acquisition_func, lower, upper, rng=rng)
    elif maximizer == "differential_evolution":
        max_func = DifferentialEvolution(acquisition_func, lower, upper, rng=rng)
    else:
        # (the reference prints an error and returns None here, entropy_search.py:110-112; a ValueError is kinder)
        raise ValueError("%s is not a valid function to maximize the acquisition function" % maximizer)
    return gp, acquisition_func, max_func


def entropy_search(objective_function, lower, upper, num_iterations=30, maximizer="random", model="gp_mcmc",
                   X_init=None, Y_init=None, n_init=3, output_path=None, rng=None, n_candidates=500,
                   chain_length=200, burnin_steps=100, n_representer=50, n_outcomes=400, n_gpus=None, devices=None):
    """``n_gpus`` / ``devices``: single-process multi-GPU (see robo_amd.fmin.bayesian_optimization): ``model="gp"`` splits the
    candidate batch of the information gain over replicas of the model, ``"gp_mcmc"`` splits the hyper-parameter samples --
    each sample's estimator (representer points, EP, gains) works on its sample's device, all devices at once."""
    assert upper.shape[0] == lower.shape[0], "Dimension miss match"
    assert np.all(lower < upper), "Lower bound >= upper bound"
    assert n_init <= num_iterations, "Number of initial design point has to be <= than the number of iterations"
    if rng is None:
        rng = np.random.RandomState(np.random.randint(0, 10000))
    from robo_amd import _lib
    gp, acquisition_func, max_func = build_entropy_search(lower, upper, maximizer, model, rng, n_candidates,
                                                          chain_length, burnin_steps, n_representer, n_outcomes,
                                                          devices=_lib.resolve_devices(devices, n_gpus))
    bo = BayesianOptimization(objective_function, lower, upper, acquisition_func, gp, max_func,
                              initial_design=init_latin_hypercube_sampling, initial_points=n_init, rng=rng,
                              output_path=output_path)
    x_best, f_min = bo.run(num_iterations, X=X_init, y=Y_init)
    return {"x_opt": x_best, "f_opt": f_min, "incumbents": [inc for inc in bo.incumbents],
            "incumbent_values": [val for val in bo.incumbents_values], "runtime": bo.runtime,
            "overhead": bo.time_overhead, "X": [x.tolist() for x in bo.X], "y": [y for y in bo.y]}

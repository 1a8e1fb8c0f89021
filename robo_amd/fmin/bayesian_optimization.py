"""``bayesian_optimization`` front end with the signature and defaults of
robo/fmin/bayesian_optimization.py:27-158 (the reference module cannot even be imported
without george, pybnn and pyrfr: :2,5,8,11), wired to the MI355X GP path:

    kernel   2 * Matern52Kernel(ones(D), ndim=D)            (:75-81)
    prior    DefaultPrior(len(kernel) + 1)                   (:83)
    n_hypers 3 * len(kernel), made even                      (:85-87)
    model    GaussianProcess | GaussianProcessMCMC(chain_length=200, burnin_steps=100)  (:89-100)
    acq      EI | LogEI | PI | LCB, wrapped in MarginalizationGPMCMC for gp_mcmc       (:114-129)

    maximiser RandomSampling | SciPyOptimizer | DifferentialEvolution                    (:131-139)

Model types other than the two GP ones (rf / bohamiann / dngo) are outside this project's hot path
(SURVEY.md section 2, rows 6, 8).
"""
import logging

import numpy as np

from robo_amd.acquisition_functions import EI, LCB, PI, LogEI, MarginalizationGPMCMC
from robo_amd.initial_design import init_latin_hypercube_sampling
from robo_amd.kernels import Matern52Kernel
from robo_amd.maximizers import DeviceRandomSampling, DifferentialEvolution, RandomSampling, SciPyOptimizer
from robo_amd.models import GaussianProcess, GaussianProcessMCMC
from robo_amd.priors import DefaultPrior
from robo_amd.solver import BayesianOptimization

logger = logging.getLogger(__name__)


def bayesian_optimization(objective_function, lower, upper, num_iterations=30, X_init=None, Y_init=None,
                          maximizer="random", acquisition_func="log_ei", model_type="gp_mcmc", n_init=3, rng=None,
                          output_path=None, n_candidates=500, chain_length=200, burnin_steps=100, n_gpus=None,
                          devices=None):
    """Minimise ``objective_function`` over the box [lower, upper] -> dict with x_opt, f_opt,
    incumbents, incumbent_values, runtime, overhead, X, y (same keys as the reference).

    ``n_candidates`` (default 500 = the reference's RandomSampling.n_samples) may be raised by
    orders of magnitude: the candidate batch is evaluated by one device call.

    ``n_gpus=G`` (devices 0 .. G-1) or ``devices=[...]``: single-process multi-GPU -- this one process drives all the
    listed devices and the objective is evaluated ONCE per iteration, exactly as in the reference's loop
    (robo/solver/bayesian_optimization.py:156-203).  ``model_type="gp"``: the fitted model is replicated on every
    device and the candidate batch of each maximisation is split over them; ``"gp_mcmc"``: the hyper-parameter samples
    (and, for large N, the walkers of the chain) are split over the devices.  Same chosen points as on one device.
    """
    assert upper.shape[0] == lower.shape[0], "Dimension miss match"
    assert np.all(lower < upper), "Lower bound >= upper bound"
    assert n_init <= num_iterations, "Number of initial design point has to be <= than the number of iterations"
    if rng is None:
        rng = np.random.RandomState(np.random.randint(0, 10000))

    cov_amp = 2
    n_dims = lower.shape[0]
    kernel = cov_amp * Matern52Kernel(np.ones([n_dims]), ndim=n_dims)
    prior = DefaultPrior(len(kernel) + 1)
    n_hypers = 3 * len(kernel)
    if n_hypers % 2 == 1:
        n_hypers += 1

    from robo_amd import _lib
    devices = _lib.resolve_devices(devices, n_gpus)
    if model_type == "gp":
        model = GaussianProcess(kernel, prior=prior, rng=rng, normalize_output=False, normalize_input=True,
                                lower=lower, upper=upper, devices=devices)
    elif model_type == "gp_mcmc":
        model = GaussianProcessMCMC(kernel, prior=prior, n_hypers=n_hypers, chain_length=chain_length,
                                    burnin_steps=burnin_steps, normalize_input=True, normalize_output=False,
                                    rng=rng, lower=lower, upper=upper, devices=devices)
    else:
        raise ValueError("'{}' is not a valid model (robo_amd provides 'gp' and 'gp_mcmc')".format(model_type))

    acq_classes = {"ei": EI, "log_ei": LogEI, "pi": PI, "lcb": LCB}
    if acquisition_func not in acq_classes:
        raise ValueError("'{}' is not a valid acquisition function".format(acquisition_func))
    a = acq_classes[acquisition_func](model)
    acq = MarginalizationGPMCMC(a) if model_type == "gp_mcmc" else a

    if maximizer == "random":
        max_func = RandomSampling(acq, lower, upper, n_samples=n_candidates, rng=rng)
    elif maximizer == "device_random":
        # same recipe, candidates generated and scored on the device, only x* comes back
        max_func = DeviceRandomSampling(acq, lower, upper, n_samples=n_candidates, rng=rng)
    elif maximizer == "scipy":
        max_func = SciPyOptimizer(acq, lower, upper, rng=rng)
    elif maximizer == "differential_evolution":
        max_func = DifferentialEvolution(acq, lower, upper, rng=rng)
    else:
        raise ValueError("'{}' is not a valid function to maximize the acquisition function".format(maximizer))

    bo = BayesianOptimization(objective_function, lower, upper, acq, model, max_func, initial_points=n_init, rng=rng,
                              initial_design=init_latin_hypercube_sampling, output_path=output_path)
    x_best, f_min = bo.run(num_iterations, X=X_init, y=Y_init)

    return {"x_opt": x_best, "f_opt": f_min,
            "incumbents": [inc for inc in bo.incumbents],
            "incumbent_values": [val for val in bo.incumbents_values],
            "runtime": bo.runtime, "overhead": bo.time_overhead,
            "X": [x.tolist() for x in bo.X], "y": [y for y in bo.y]}

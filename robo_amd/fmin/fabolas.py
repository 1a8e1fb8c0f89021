"""``fabolas`` front end (Klein et al. 2016) with the signature and loop of
robo/fmin/fabolas.py:31-312: two Fabolas GP-MCMC models (log loss with basis (1-s)^2, log cost
with basis s), information gain per unit cost marginalised over the hyper-parameter samples,
RandomSampling over the (D+1)-dimensional box, incumbent by projection to the full data set."""
import json
import logging
import os
import time

import numpy as np

from robo_amd.acquisition_functions import EI, InformationGainPerUnitCost, MarginalizationGPMCMC
from robo_amd.initial_design import init_latin_hypercube_sampling
from robo_amd.kernels import FabolasKernel
from robo_amd.maximizers import RandomSampling
from robo_amd.models import FabolasGPMCMC
from robo_amd.priors import EnvPrior
from robo_amd.util.incumbent_estimation import projected_incumbent_estimation

logger = logging.getLogger(__name__)


def transform(s, s_min, s_max):
    return (np.log2(s) - np.log2(s_min)) / (np.log2(s_max) - np.log2(s_min))


def retransform(s_transform, s_min, s_max):
    return int(np.rint(2 ** (s_transform * (np.log2(s_max) - np.log2(s_min)) + np.log2(s_min))))


def build_fabolas(lower, upper, burnin=100, chain_length=100, n_hypers=12, rng=None, n_candidates=500,
                  n_representer=50, n_outcomes=400, devices=None):
    """the objects robo/fmin/fabolas.py:99-199 wires together -> (objective model, cost model, acquisition
    function, maximiser)"""
    n_dims = lower.shape[0]
    kernel = FabolasKernel(n_dims + 1, metric=0.01, log_a=0.1, log_b=0.1, amp=1.0)
    if n_hypers < 2 * len(kernel):
        n_hypers = 3 * len(kernel)
        if n_hypers % 2 == 1:
            n_hypers += 1
    prior = EnvPrior(len(kernel) + 1, n_ls=n_dims, n_lr=2, rng=rng)
    model_objective = FabolasGPMCMC(kernel, prior=prior, burnin_steps=burnin, chain_length=chain_length,
                                    n_hypers=n_hypers, normalize_output=False, basis_func=lambda s: (1 - s) ** 2,
                                    lower=lower, upper=upper, rng=rng, devices=devices)
    cost_kernel = FabolasKernel(n_dims + 1, metric=0.01, log_a=0.1, log_b=0.1, amp=1.0)
    cost_prior = EnvPrior(len(cost_kernel) + 1, n_ls=n_dims, n_lr=2, rng=rng)
    model_cost = FabolasGPMCMC(cost_kernel, prior=cost_prior, burnin_steps=burnin, chain_length=chain_length,
                               n_hypers=n_hypers, basis_func=lambda s: s, normalize_output=False, lower=lower,
                               upper=upper, rng=rng, devices=devices)
    extend_lower, extend_upper = np.append(lower, 0), np.append(upper, 1)
    is_env = np.zeros(extend_lower.shape[0])
    is_env[-1] = 1
    ig = InformationGainPerUnitCost(model_objective, model_cost, extend_lower, extend_upper, sampling_acquisition=EI,
                                    is_env_variable=is_env, n_representer=n_representer, Np=n_outcomes, rng=rng)
    acquisition_func = MarginalizationGPMCMC(ig)
    maximizer = RandomSampling(acquisition_func, extend_lower, extend_upper, n_samples=n_candidates)
    return model_objective, model_cost, acquisition_func, maximizer


def fabolas(objective_function, lower, upper, s_min, s_max, n_init=40, num_iterations=100, subsets=[256, 128, 64],
            inc_estimation="mean", burnin=100, chain_length=100, n_hypers=12, output_path=None, rng=None,
            n_candidates=500, n_representer=50, n_outcomes=400, n_gpus=None, devices=None):
    """objective_function(x, s) -> (validation error, cost); returns the reference's result dict.

    ``n_gpus`` / ``devices``: single-process multi-GPU -- the hyper-parameter samples of both models are split over the
    listed devices (sample s of the loss model and sample s of the cost model share a device); every sample's
    information gain per unit cost is evaluated on its device, all devices at once; one objective evaluation per
    iteration as in robo/fmin/fabolas.py:222-296."""
    time_start = time.time()
    if rng is None:
        rng = np.random.RandomState(np.random.randint(0, 10000))
    n_dims = lower.shape[0]
    time_func_eval, time_overhead, incumbents, runtime = [], [], [], []
    X, y, c = [], [], []

    def _dump(it):
        if output_path is not None:
            data = {"optimization_overhead": time_overhead[it], "runtime": runtime[it],
                    "incumbent": np.asarray(incumbents[it]).tolist(), "time_func_eval": time_func_eval[it],
                    "iteration": it}
            with open(os.path.join(output_path, "fabolas_iter_%d.json" % it), "w") as fh:
                json.dump(data, fh)

    from robo_amd import _lib
    model_objective, model_cost, acquisition_func, maximizer = build_fabolas(
        lower, upper, burnin, chain_length, n_hypers, rng, n_candidates, n_representer, n_outcomes,
        devices=_lib.resolve_devices(devices, n_gpus))

    x_init = init_latin_hypercube_sampling(lower, upper, n_init, rng)
    for it in range(n_init):
        for subset in subsets:
            t0 = time.time()
            s = int(s_max / float(subset))
            x = x_init[it]
            st = time.time()
            func_val, cost = objective_function(x, s)
            time_func_eval.append(time.time() - st)
            X.append(np.append(x, transform(s, s_min, s_max)))
            y.append(np.log(func_val))     # loss and cost are modelled on a log scale
            c.append(np.log(cost))
            incumbents.append(X[int(np.argmin(y))][:-1])
            time_overhead.append(time.time() - t0)
            runtime.append(time.time() - time_start)
            _dump(it)
    X, y, c = np.array(X), np.array(y), np.array(c)

    for it in range(X.shape[0], num_iterations):
        t0 = time.time()
        model_objective.train(X, y, do_optimize=True)
        model_cost.train(X, c, do_optimize=True)
        if inc_estimation == "last_seen":
            best = int(np.argmin(y))
            incumbent, incumbent_value = np.append(X[best][:-1], 1), y[best]
        else:
            incumbent, incumbent_value = projected_incumbent_estimation(model_objective, X[:, :-1], proj_value=1)
        incumbents.append(incumbent[:-1])
        logger.info("Current incumbent %s with estimated performance %f", str(incumbent), np.exp(incumbent_value))
        acquisition_func.update(model_objective, model_cost)
        new_x = maximizer.maximize()
        s = retransform(new_x[-1], s_min, s_max)
        time_overhead.append(time.time() - t0)
        t0 = time.time()
        new_y, new_c = objective_function(new_x[:-1], s)
        time_func_eval.append(time.time() - t0)
        X = np.concatenate((X, new_x[None, :]), axis=0)
        y = np.concatenate((y, np.log(np.array([new_y]))), axis=0)
        c = np.concatenate((c, np.log(np.array([new_c]))), axis=0)
        runtime.append(time.time() - time_start)
        _dump(it)

    model_objective.train(X, y, do_optimize=True)
    incumbent, incumbent_value = projected_incumbent_estimation(model_objective, X[:, :-1], proj_value=1)
    return {"x_opt": incumbent[:-1].tolist(), "incumbents": [np.asarray(inc).tolist() for inc in incumbents],
            "runtime": runtime, "overhead": time_overhead, "time_func_eval": time_func_eval,
            "X": [x.tolist() for x in X], "y": [np.exp(yi).tolist() for yi in y], "c": [ci.tolist() for ci in c]}

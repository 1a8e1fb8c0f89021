"""Incumbent estimation by minimising the posterior mean (or mean + std) with multi-start L-BFGS-B.

Signatures and semantics of robo/util/posterior_optimization.py:8-118.  ``with_gradients=True`` is the
consumer of ``model.predictive_gradients`` there (:38-40, :96-104) -- a path no model of the reference can
serve; robo_amd's GaussianProcess does (device side: robo_gp_predict_grad), so both variants run.
Host code: a handful of (1, D) posterior evaluations per line-search step.
"""
import numpy as np
from scipy import optimize

from robo_amd.initial_design import init_random_uniform


def _multi_start(f, df, lower, upper, n_restarts, with_gradients):
    starts = init_random_uniform(lower, upper, n_restarts)
    x_opt = np.zeros([len(starts), lower.shape[0]])
    fval = np.zeros([len(starts)])
    bounds = list(zip(lower, upper))
    for i, x0 in enumerate(starts):
        if with_gradients:
            res = optimize.fmin_l_bfgs_b(f, x0, df, bounds=bounds)
            x_opt[i], fval[i] = res[0], res[1]
        else:
            res = optimize.minimize(f, x0, bounds=bounds, method="L-BFGS-B")
            x_opt[i], fval[i] = res["x"], res["fun"]
    return x_opt[np.argmin(fval)]


def posterior_mean_optimization(model, lower, upper, n_restarts=10, with_gradients=False):
    """argmin of the posterior mean over the box -> (D,)"""

    def f(x):
        return model.predict(x[np.newaxis, :])[0][0]

    def df(x):
        return np.asarray(model.predictive_gradients(x[np.newaxis, :])[0]).reshape(-1)

    return _multi_start(f, df, lower, upper, n_restarts, with_gradients)


def posterior_mean_plus_std_optimization(model, lower, upper, n_restarts=10, with_gradients=False):
    """argmin of posterior mean + standard deviation (the upper confidence bound) over the box -> (D,)"""

    def f(x):
        mu, var = model.predict(x[np.newaxis, :])
        return (mu + np.sqrt(var))[0]

    def df(x):
        dmu, dvar = model.predictive_gradients(x[np.newaxis, :])
        _, var = model.predict(x[np.newaxis, :])
        # s = sqrt(v)  =>  ds/dx = v'(x) / (2 sqrt(v))
        return (np.asarray(dmu)[:, :, 0] + 0.5 * np.asarray(dvar) / np.sqrt(var)[:, None]).reshape(-1)

    return _multi_start(f, df, lower, upper, n_restarts, with_gradients)

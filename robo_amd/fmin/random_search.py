"""Random search, the model-free baseline next to the three model-based front ends (robo/fmin/random_search.py:11-121):
``num_iterations`` points drawn uniformly in the box with ``rng.uniform(lower, upper)`` -- one D-vector per iteration, so a
seeded run visits the reference's points --, the best observation so far as the incumbent, the reference's result keys and
per-iteration ``robo_iter_%d.json``.  Nothing here touches the device.

One difference, on purpose: the reference appends to its mutable default arguments ``X_init=[]`` / ``Y_init=[]``, so a second
call in the same process starts from the first call's points; here the given lists are copied.
"""
import json
import logging
import os
import time

import numpy as np

logger = logging.getLogger(__name__)


def random_search(objective_function, lower, upper, X_init=[], Y_init=[], num_iterations=30, output_path=None, rng=None):
    t_begin = time.time()
    rng = np.random.RandomState() if rng is None else rng
    X, y = list(X_init), list(Y_init)
    log = {k: [] for k in ("incumbents", "incumbent_values", "runtime", "overhead", "time_func_eval")}
    for it in range(num_iterations):
        t0 = time.time()
        x = rng.uniform(lower, upper)
        if lower.shape[0] == 1:
            x = np.array([x])       # as the reference (:66-68): with array bounds of length 1 the point becomes (1, 1)
        log["overhead"].append(time.time() - t0)
        t0 = time.time()
        value = objective_function(x)
        log["time_func_eval"].append(time.time() - t0)
        logger.info("Iteration %d: %s -> %f", it, x, value)
        X.append(x.tolist())
        y.append(value)
        best = int(np.argmin(y))
        log["incumbents"].append(X[best])
        log["incumbent_values"].append(y[best])
        log["runtime"].append(time.time() - t_begin)
        if output_path is not None:
            record = {"optimization_overhead": log["overhead"][it], "runtime": log["runtime"][it],
                      "incumbent": log["incumbents"][it], "incumbents_value": log["incumbent_values"][it],
                      "time_func_eval": log["time_func_eval"][it], "iteration": it}
            with open(os.path.join(output_path, "robo_iter_%d.json" % it), "w") as fh:
                json.dump(record, fh)
    results = dict(log, X=X, y=y)
    results["x_opt"], results["f_opt"] = log["incumbents"][-1], log["incumbent_values"][-1]
    return results

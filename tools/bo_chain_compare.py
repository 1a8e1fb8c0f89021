import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from robo_amd.fmin import bayesian_optimization
def branin(x):
    return float((x[1] - 5.1 / (4 * np.pi ** 2) * x[0] ** 2 + 5 / np.pi * x[0] - 6) ** 2 + 10 * (1 - 1 / (8 * np.pi)) * np.cos(x[0]) + 10)
res = {}
for mode in ("0", "1"):
    os.environ["ROBO_MCMC_HOST"] = mode
    np.random.seed(0)      # DefaultPrior() without an rng seeds itself from the global stream (as in the reference)
    t0 = time.time()
    r = bayesian_optimization(branin, np.array([-5.0, 0.0]), np.array([10.0, 15.0]), num_iterations=60, n_init=3,
                              model_type="gp_mcmc", acquisition_func="log_ei", maximizer="random", rng=np.random.RandomState(3))
    res[mode] = r
    print("%s: 60 iterations %.2f s, f_opt %.6f (regret %.2e), mean overhead per iteration %.1f ms" %
          ("host sampler" if mode == "1" else "device chain", time.time() - t0, r["f_opt"], r["f_opt"] - 0.397887, 1e3 * np.mean(r["overhead"][3:])), flush=True)
Xd, Xh = np.array(res["0"]["X"]), np.array(res["1"]["X"])
same = int(np.sum(np.all(np.isclose(Xd, Xh, rtol=0, atol=1e-9), axis=1)))
print("evaluated points identical in %d of %d iterations (first difference at %s)" % (same, len(Xd), next((i for i in range(len(Xd)) if not np.allclose(Xd[i], Xh[i], rtol=0, atol=1e-9)), None)))

import sys, os, time, cProfile, pstats, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from robo_amd.fmin import bayesian_optimization
D = 6
def f(x):
    return float(np.sum((x - 0.3) ** 2) + 0.1 * np.sin(10 * x[0]))
lo, hi = np.zeros(D), np.ones(D)
rng = np.random.RandomState(0)
X0 = rng.rand(40, D); Y0 = np.array([f(x) for x in X0])
t = time.time()
pr = cProfile.Profile(); pr.enable()
r = bayesian_optimization(f, lo, hi, num_iterations=45, X_init=X0, Y_init=Y0, n_init=40, model_type="gp_mcmc",
                          acquisition_func="log_ei", rng=np.random.RandomState(1))
pr.disable()
print("5 BO iterations (N=40..44, D=6, gp_mcmc defaults: 24 walkers, burnin 100 + 5x chain 200): %.2f s, overhead/iter %s" %
      (time.time() - t, np.round(r["overhead"][-5:], 3)))
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)

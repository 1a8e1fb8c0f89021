"""latency of the analytic likelihood gradient (MAP / L-BFGS path) at BO-typical N"""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from robo_amd import _lib
ctx = _lib.Context(0)
for N, D in ((30, 2), (100, 6), (300, 8), (1000, 16), (4096, 16)):
    X = np.random.RandomState(0).rand(N, D); y = np.sin(X.sum(axis=1))
    theta = np.concatenate([[0.0], np.full(D, np.log(0.25 * D)), [np.log(1e-3)]])
    g = _lib.DeviceGP(ctx, "matern52", N, D); g.set_data(X, y)
    for _ in range(3): g.grad_loglik(theta, 0.0)
    reps = 30 if N <= 1000 else 5
    t0 = time.perf_counter()
    for _ in range(reps): g.grad_loglik(theta, 0.0)
    tg = (time.perf_counter() - t0) / reps * 1e3
    t0 = time.perf_counter()
    for _ in range(reps): g.fit(theta, 0.0)
    tf = (time.perf_counter() - t0) / reps * 1e3
    print("N=%5d D=%2d  loglik + gradient %.3f ms   (fit alone %.3f ms)" % (N, D, tg, tf))
    g.close()

"""latency of the fit paths at BO-typical small N (where launches, not flops, are the cost)"""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from robo_amd import _lib
ctx = _lib.Context(0)
for N, D in ((30, 2), (100, 6), (300, 8), (1000, 16)):
    X = np.random.RandomState(0).rand(N, D); y = np.sin(X.sum(axis=1))
    theta = np.concatenate([[0.0], np.full(D, np.log(0.25 * D)), [np.log(1e-3)]])
    g = _lib.DeviceGP(ctx, "matern52", N, D); g.set_data(X, y)
    S = (3 * (D + 2) + (3 * (D + 2)) % 2) // 2
    thetas = theta[None, :] + 0.1 * np.random.RandomState(1).randn(S, theta.size)
    for _ in range(3): g.fit(theta, 0.0); g.loglik_batch(thetas, 0.0)
    t0 = time.perf_counter()
    for _ in range(50): g.fit(theta, 0.0)
    t_fit = (time.perf_counter() - t0) / 50 * 1e3
    t0 = time.perf_counter()
    for _ in range(50): g.loglik_batch(thetas, 0.0)
    t_b = (time.perf_counter() - t0) / 50 * 1e3
    Xc = np.random.RandomState(2).rand(500, D); c = _lib.Candidates(ctx, Xc)
    g.fit(theta, 0.0)
    for _ in range(3): g.acq("ei", 0.0, 0.0, c, want_values=False)
    t0 = time.perf_counter()
    for _ in range(50): g.acq("ei", 0.0, 0.0, c, want_values=False)
    t_a = (time.perf_counter() - t0) / 50 * 1e3
    t0 = time.perf_counter()
    for _ in range(50): g.acq("ei", 0.0, 0.0, Xc, want_values=True)
    t_h = (time.perf_counter() - t0) / 50 * 1e3
    print("N=%4d D=%2d  fit %.3f ms   loglik_batch(S=%d) %.3f ms (%.3f/theta)   EI(500 resident) %.3f ms   EI(500 from host) %.3f ms"
          % (N, D, t_fit, S, t_b, t_b / S, t_a, t_h))

import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from robo_amd import _lib
ctx = _lib.Context(0)
for N, D, S in ((100, 8, 26), (500, 8, 26), (2048, 16, 26), (4096, 16, 26)):
    rs = np.random.RandomState(0)
    X = rs.rand(N, D); y = np.sin(X.sum(axis=1))
    base = np.concatenate([[0.0], np.full(D, np.log(0.25 * D)), [np.log(1e-3)]])
    thetas = base[None, :] + 0.2 * rs.randn(S, base.size)
    g = _lib.DeviceGP(ctx, "matern52", N, D); g.set_data(X, y)
    g.loglik_batch(thetas, 0.0)
    t0 = time.perf_counter(); 
    for _ in range(3): ll, st = g.loglik_batch(thetas, 0.0)
    tb = (time.perf_counter() - t0) / 3
    g.fit(thetas[0], 0.0)
    t0 = time.perf_counter()
    for s in range(S): g.fit(thetas[s], 0.0)
    ts = time.perf_counter() - t0
    print("N=%5d D=%2d S=%d: batched %.3f ms (%.3f ms/theta)  sequential %.3f ms (%.3f ms/theta)  speedup %.1fx" %
          (N, D, S, tb * 1e3, tb * 1e3 / S, ts * 1e3, ts * 1e3 / S, ts / tb))
    g.close()

"""posterior + EI latency for 1 and 8 candidates (the reference's single-point maximisers call with 1 x D): matrix-vector form of
the explicit inverse vs the other forms, over training-set sizes"""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from robo_amd import _lib
ctx = _lib.Context(0)
def best(f, reps=20):
    f(); ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); f(); ts.append((time.perf_counter() - t0) * 1e3)
    return min(ts)
for N, D in ((200, 8), (300, 8), (500, 8), (1000, 8), (2048, 16), (4096, 16)):
    X = np.random.RandomState(0).rand(N, D); y = np.sinc(X * 10 - 5).sum(axis=1); y = (y - y.mean()) / y.std()
    theta = np.concatenate([[0.0], np.full(D, np.log(0.25 * D)), [np.log(1e-3)]])
    g = _lib.DeviceGP(ctx, "matern52", N, D); g.set_data(X, y); g.fit(theta, 0.0)
    eta = float(y.min())
    for M in (1, 8):
        cand = _lib.Candidates(ctx, np.random.RandomState(1).rand(M, D))
        res = []
        for label, sets in (("default", {}), ("gemv", {"winv_min_blocks": 1, "winv_gemv": 1}), ("chunked", {"winv_min_blocks": 1, "winv_gemv": 0}),
                            ("substitution", {"winv_max": 0})):
            for k, v in sets.items(): ctx.set_tuning(k, v)
            res.append((label, best(lambda: g.acq("ei", 0.0, eta, cand, want_values=False)), cand.solve_kernel()))
            for k in sets: ctx.set_tuning(k, None)
        print("N=%5d M=%d: " % (N, M) + "  ".join("%s %.3f ms (%s)" % r for r in res))
        cand.close()
    g.close()

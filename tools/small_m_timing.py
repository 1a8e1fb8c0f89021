"""posterior + EI latency for small candidate batches at the headline GP (N=4096, D=16): 32- vs 128-candidate step"""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from robo_amd import _lib
N, D = 4096, 16
ctx = _lib.Context(0)
X = np.random.RandomState(0).rand(N, D); y = np.sinc(X * 10 - 5).sum(axis=1); y = (y - y.mean()) / y.std()
theta = np.concatenate([[0.0], np.full(D, np.log(0.25 * D)), [np.log(1e-3)]])
g = _lib.DeviceGP(ctx, "matern52", N, D); g.set_data(X, y); g.fit(theta, 0.0)
for M in (128, 500, 2048, 8192, 16384, 32768):
    cand = _lib.Candidates(ctx, np.random.RandomState(1).rand(M, D))
    out = []
    for small in ("0", "1000000"):
        os.environ["ROBO_TRSM_SMALL_MAX"] = small
        g.acq("ei", 0.0, float(y.min()), cand, want_values=False)
        ts = []
        for _ in range(5):
            t0 = time.perf_counter(); g.acq("ei", 0.0, float(y.min()), cand, want_values=False); ts.append((time.perf_counter() - t0) * 1e3)
        out.append(min(ts))
    print("M=%6d: 128-candidate step %.3f ms, 32-candidate step %.3f ms" % (M, out[0], out[1]))

"""N fits at the headline size (for rocprofv3 --kernel-trace --stats of the fit kernels alone)."""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from robo_amd import _lib
N, D = int(os.environ.get("FIT_N", 4096)), int(os.environ.get("FIT_D", 16))
ctx = _lib.Context(0)
X = np.random.RandomState(0).rand(N, D); y = np.sinc(X * 10 - 5).sum(axis=1); y = (y - y.mean()) / y.std()
theta = np.concatenate([[0.0], np.full(D, np.log(0.25 * D)), [np.log(1e-3)]])
g = _lib.DeviceGP(ctx, "matern52", N, D); g.set_data(X, y)
ts = []
for _ in range(int(os.environ.get("FIT_REPS", 10))):
    t0 = time.perf_counter(); g.fit(theta, float(y.mean())); ts.append((time.perf_counter() - t0) * 1e3)
print("fit ms: min %.3f median %.3f" % (min(ts), sorted(ts)[len(ts) // 2]))

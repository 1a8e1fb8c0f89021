import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from robo_amd import _lib
if len(sys.argv) > 1:
    _lib.use_library(os.path.abspath(sys.argv[1]))
    print("== library", sys.argv[1])
ctx = _lib.Context(0)
N, D = 1000, 8
X = np.random.RandomState(0).rand(N, D); y = np.sin(X.sum(axis=1))
theta = np.concatenate([[0.0], np.full(D, np.log(0.25 * D)), [np.log(1e-3)]])
g = _lib.DeviceGP(ctx, "matern52", N, D); g.set_data(X, y)
for _ in range(3):
    t = g.diag_timeline(theta)
names = ["start", "load", "potf2(0)", "subpanel(0)"] + ["step%d" % i for i in range(7)] + ["inverse", "writeback"]
prev = 0
for n, v in zip(names, t[:13]):
    print("%-12s %8.0f cycles  (+%6.0f)" % (n, v, v - prev)); prev = v
prev = 0
for n, v in zip(["panel start", "operands in LDS", "144-MFMA chain", "stored"], t[13:17]):
    print("%-16s %8.0f cycles  (+%6.0f)" % (n, v, v - prev)); prev = v

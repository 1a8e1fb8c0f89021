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
print("pivot wave per interval s (cycles since kernel start): C1+C2 done | through Bb | potf2(s+1) done")
ba = [t[3 - 1]] + list(t[4:11])        # Ba(0) = stamp 2 ("potf2(0)"), Ba(s+1) = stamps 4+s
for sb in range(7):
    c, bb, pf = t[17 + 3 * sb: 20 + 3 * sb]
    print("s=%d  Ba->C done %5.0f   wait at Bb %5.0f   potf2 %5.0f   wait at Ba %5.0f" % (sb, c - ba[sb], bb - c, pf - bb, ba[sb + 1] - pf))

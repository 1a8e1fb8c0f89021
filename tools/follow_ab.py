"""Same-session A/B of the single-theta factorisation: launch-per-phase (potrf_follow = 0) against the follower form
(potrf_follow = 1) started at several steps; median / min of REPS data-resident fits, bits of the likelihood compared.

    python tools/follow_ab.py [N] [D] [reps]        # FOLLOW_FROM="-1,0,3,5,8,11"
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from robo_amd import _lib  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
D = int(sys.argv[2]) if len(sys.argv) > 2 else 16
REPS = int(sys.argv[3]) if len(sys.argv) > 3 else 15
froms = [int(v) for v in os.environ.get("FOLLOW_FROM", "-1,0,3,5,8,11").split(",")]
ctx = _lib.Context(0)
X = np.random.RandomState(0).rand(N, D)
y = np.sinc(X * 10 - 5).sum(axis=1)
y = (y - y.mean()) / y.std()
theta = np.concatenate([[0.0], np.full(D, np.log(0.25 * D)), [np.log(1e-3)]])
g = _lib.DeviceGP(ctx, "matern52", N, D)
g.set_data(X, y)
ref = None
print("single-theta fit, N=%d D=%d, %d reps each (ms: median / min)" % (N, D, REPS))
early = [int(v) for v in os.environ.get("FOLLOW_EARLY", "6").split(",")]
rows = [int(v) for v in os.environ.get("FOLLOW_ROWS", "-1").split(",")]
sleeps = [int(v) for v in os.environ.get("FOLLOW_SLEEP", "1").split(",")]
for rnd in range(2):
    for follow, frm, ea, fr, sl in [(0, 0, 6, 64, 1)] + [(1, f, e, r, z) for f in froms for e in early for r in rows for z in sleeps]:
        ctx.set_tuning("potrf_poll_sleep", sl)
        ctx.set_tuning("potrf_pub_early", ea)
        ctx.set_tuning("potrf_follow_rows", fr)
        ctx.set_tuning("potrf_follow", follow)
        ctx.set_tuning("potrf_follow_from", frm)
        ll = g.fit(theta, 0.0)
        if ref is None:
            ref = ll
        ts = []
        for _ in range(REPS):
            t0 = time.perf_counter()
            g.fit(theta, 0.0)
            ts.append((time.perf_counter() - t0) * 1e3)
        print("round %d  follow %d from %2d early %d rows %3d sleep %2d : %.4f / %.4f ms   bits %s" % (
            rnd, follow, frm, ea, fr, sl, sorted(ts)[len(ts) // 2], min(ts), "same" if ll == ref else "DIFFER (%r vs %r)" % (ll, ref)), flush=True)
g.close()

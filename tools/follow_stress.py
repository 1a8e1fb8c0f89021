"""Hand-off stress: fits in the follower form (single-theta step kernel and the batched merged launch) while ANOTHER context
of the same device keeps the chip busy with large posterior evaluations (uneven load, other kernels in the CUs' L1 / L2) --
every word of every factor and every likelihood must equal the launch-per-phase reference, iteration after iteration.

    python tools/follow_stress.py [iterations]
"""
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from robo_amd import _lib  # noqa: E402

ITERS = int(sys.argv[1]) if len(sys.argv) > 1 else 150
ctx, ctx2 = _lib.Context(0), _lib.Context(0)
stop = threading.Event()


def background():
    N, D, M = 2048, 8, 32768
    rs = np.random.RandomState(9)
    X = rs.rand(N, D)
    y = np.sin(X.sum(axis=1))
    th = np.concatenate([[0.0], np.full(D, np.log(0.25 * D)), [np.log(1e-3)]])
    g = _lib.DeviceGP(ctx2, "matern52", N, D)
    g.set_data(X, y)
    g.fit(th, 0.0)
    cand = _lib.Candidates(ctx2, rs.rand(M, D))
    n = 0
    while not stop.is_set():
        g.acq("ei", 0.0, float(y.min()), cand, want_values=False)
        n += 1
    background.count = n


bad = 0
t = threading.Thread(target=background)
t.start()
try:
    for N, D in ((4096, 16), (1500, 6), (700, 3)):
        rs = np.random.RandomState(N)
        X = rs.rand(N, D)
        y = np.sinc(X * 10 - 5).sum(axis=1)
        th = np.concatenate([[0.0], np.full(D, np.log(0.25 * D)), [np.log(1e-3)]])
        g = _lib.DeviceGP(ctx, "matern52", N, D)
        g.set_data(X, y)
        ctx.set_tuning("potrf_follow", 0)
        ll0 = g.fit(th, 0.0)
        L0 = g.factor().copy()
        ctx.set_tuning("potrf_follow", None)
        S = 12
        thetas = th[None, :] + 0.1 * rs.randn(S, th.size)
        ctx.set_tuning("potrf_batch_follow", 0)
        b0, _ = g.loglik_batch(thetas, 0.0)
        t0 = time.perf_counter()
        for it in range(ITERS):
            ctx.set_tuning("potrf_follow_rows", (64, 128, -1)[it % 3])
            ll = g.fit(th, 0.0)
            L = g.factor() if it % 10 == 0 else None
            if ll != ll0 or (L is not None and not np.array_equal(L, L0)):
                bad += 1
                print("MISMATCH single N=%d it=%d  dll=%g" % (N, it, ll - ll0), flush=True)
            ctx.set_tuning("potrf_batch_follow", 1)
            ctx.set_tuning("potrf_batch_roll", it & 1)
            b, st = g.loglik_batch(thetas, 0.0)
            if not np.array_equal(b, b0):
                bad += 1
                print("MISMATCH batched N=%d it=%d  max d=%g" % (N, it, np.abs(b - b0).max()), flush=True)
        for k in ("potrf_follow_rows", "potrf_batch_follow", "potrf_batch_roll"):
            ctx.set_tuning(k, None)
        print("N=%d: %d iterations (single-theta follower fit + batched merged launch, %d thetas) under load: %s  (%.1f s)" % (
            N, ITERS, S, "all bit-identical" if bad == 0 else "%d MISMATCHES" % bad, time.perf_counter() - t0), flush=True)
        g.close()
finally:
    stop.set()
    t.join()
print("background posterior evaluations meanwhile:", getattr(background, "count", None))
sys.exit(1 if bad else 0)

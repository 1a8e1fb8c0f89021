"""Same-session A/B of the batched factorisation's schedules (potrf_group -- 0 = by size, the default -- / potrf_split / potrf_lead):
S likelihoods of ONE robo_gp_loglik_batch call at N x D, per-theta time and fraction of the fp64 MFMA peak
(S N^3 / 3 flops), likelihoods compared bit for bit with the first variant.

    python tools/batched_fit_ab.py [N] [D] [S] [reps]      # variants: BATCH_AB="g,split,lead;g,split,lead;..."
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from robo_amd import _lib  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
D = int(sys.argv[2]) if len(sys.argv) > 2 else 16
S = int(sys.argv[3]) if len(sys.argv) > 3 else 27
REPS = int(sys.argv[4]) if len(sys.argv) > 4 else 5
spec = os.environ.get("BATCH_AB", "4,1,-1;6,1,-1;8,1,-1;12,1,-1;6,2,-1;6,2,1;6,2,2;8,2,-1;6,3,-1;8,3,-1;6,4,-1;4,2,-1")
variants = [tuple(int(v) for v in item.split(",")) for item in spec.split(";") if item]

ctx = _lib.Context(0)
for item in os.environ.get("BATCH_TUNE", "").split(","):          # e.g. BATCH_TUNE="potrf_thin_last=0"
    if "=" in item:
        key, val = item.split("=")
        ctx.set_tuning(key.strip(), int(val))
X = np.random.RandomState(0).rand(N, D)
y = np.sinc(X * 10 - 5).sum(axis=1)
y = (y - y.mean()) / y.std()
theta = np.concatenate([[0.0], np.full(D, np.log(0.25 * D)), [np.log(1e-3)]])
thetas = theta[None, :] + 0.1 * np.random.RandomState(2).randn(S, theta.size)
g = _lib.DeviceGP(ctx, "matern52", N, D)
g.set_data(X, y)
ref = None
print("batched fit: N=%d D=%d S=%d, %d reps each (median / min ms per theta, fraction of 78.6 TFLOP/s)" % (N, D, S, REPS))
for rnd in range(2):                       # two rounds: the second sees the part at its sustained clock
    for group, split, lead in variants:
        ctx.set_tuning("potrf_group", group)
        ctx.set_tuning("potrf_split", split)
        ctx.set_tuning("potrf_lead", lead)
        g.loglik_batch(thetas + 0.01, 0.0)            # other gram matrices in the workspace first (a schedule that reads K
        ll, st = g.loglik_batch(thetas, 0.0)          # too early must not find the right ones left over); warm-up
        assert np.all(st == _lib.OK), st
        if ref is None:
            ref = ll.copy()
        same = bool(np.array_equal(ll, ref))
        ts = []
        for _ in range(REPS):
            ctx.synchronize()
            t0 = time.perf_counter()
            g.loglik_batch(thetas, 0.0)
            ts.append((time.perf_counter() - t0) * 1e3)
        med, mn = sorted(ts)[len(ts) // 2], min(ts)
        frac = S * N ** 3 / 3.0 / (med * 1e-3) / 78.6e12
        print("round %d  group %2d split %d lead %2d : %.4f / %.4f ms per theta  total %.3f ms  frac %.3f  bits %s"
              % (rnd, group, split, lead, med / S, mn / S, med, frac, "same" if same else "DIFFER"), flush=True)
g.close()

"""Block-row posterior step: every second workgroup held back at the start of each launch (trsm_skew, microseconds; which
workgroups: bit trsm_skew_shift of the index) x block rows per launch (trsm_rows) -- same-session A/B, EI over M candidates,
argmax compared.       python tools/skew_ab.py N D M steps"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from robo_amd import _lib  # noqa: E402

N, D, M, STEPS = (int(v) for v in (sys.argv[1:5] + ["4096", "16", "65536", "10"][len(sys.argv) - 1:]))
variants = os.environ.get("SKEW_AB", "0,8,1;4,8,1;8,8,1;12,8,1;8,0,1;8,3,1;8,8,2;8,8,4;0,8,2;0,8,4")
ctx = _lib.Context(0)
rs = np.random.RandomState(0)
X = rs.rand(N, D)
y = np.sinc(X * 10 - 5).sum(axis=1)
y = (y - y.mean()) / y.std()
theta = np.concatenate([[0.0], np.full(D, np.log(0.25 * D)), [np.log(1e-3)]])
g = _lib.DeviceGP(ctx, "matern52", N, D)
g.set_data(X, y)
g.fit(theta, 0.0)
cand = _lib.Candidates(ctx, np.random.RandomState(1).rand(M, D))
eta = float(y.min())
ref = None
print("EI over %d candidates, N=%d D=%d, %d steps each" % (M, N, D, STEPS))
for rnd in range(2):
    for item in variants.split(";"):
        skew, shift, rows = (int(v) for v in item.split(","))
        ctx.set_tuning("trsm_skew", skew)
        ctx.set_tuning("trsm_skew_shift", shift)
        ctx.set_tuning("trsm_rows", rows)
        for _ in range(3):
            out = g.acq("ei", 0.0, eta, cand, want_values=False)
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(STEPS):
            out = g.acq("ei", 0.0, eta, cand, want_values=False)
        dt = (time.perf_counter() - t0) / STEPS
        if ref is None:
            ref = (out[1], out[2])
        print("round %d  skew %2d us shift %d rows %d : %.3f ms per step = %.3f M evals/s   argmax %s" % (
            rnd, skew, shift, rows, dt * 1e3, M / dt / 1e6, "same" if (out[1], out[2]) == ref else "DIFFERS"), flush=True)

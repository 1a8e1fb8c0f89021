"""Same-session A/B of K1 (gram_kernel) between builds of librobo_hip.so: the gram kernel alone (phase events 19 -> 21) and the
whole fit, N x D.      python tools/k1_ab.py lib_a.so lib_b.so ... [--n 4096 --d 16]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from robo_amd import _lib  # noqa: E402

args = sys.argv[1:]
N, D = 4096, 16
FP32 = False
libs = []
while args:
    a = args.pop(0)
    if a == "--n":
        N = int(args.pop(0))
    elif a == "--d":
        D = int(args.pop(0))
    elif a == "--fp32":
        FP32 = True
    else:
        libs.append(a)
X = np.random.RandomState(0).rand(N, D)
y = np.sinc(X * 10 - 5).sum(axis=1)
y = (y - y.mean()) / y.std()
theta = np.concatenate([[0.0], np.full(D, np.log(0.25 * D)), [np.log(1e-3)]])
ref = None
for rnd in range(2):
    for path in libs:
        _lib.use_library(None if path == "default" else os.path.abspath(path))
        ctx = _lib.Context(0)
        g = _lib.DeviceGP(ctx, "matern52", N, D)
        g.set_data(X, y)
        if FP32:
            g.set_precision(True)            # covariance entries evaluated in fp32, widened to fp64 (BASELINE config 5)
        ll = g.fit(theta, 0.0)
        if ref is None:
            ref = ll
        ctx.set_phase_events(True)
        k1, fit = [], []
        for _ in range(15):
            t0 = time.perf_counter()
            g.fit(theta, 0.0)
            fit.append((time.perf_counter() - t0) * 1e3)
            k1.append(ctx.elapsed_ms(19, 21) * 1e3)
        ctx.set_phase_events(False)
        nbytes = 8.0 * N * (N + 1) / 2 + 8.0 * N * D        # K is stored in fp64 in both precisions
        print("round %d  %s %-32s K1 %.1f us (min %.1f) = %.2f TB/s = %.3f of 8 TB/s;  fit %.4f ms;  loglik bits %s" % (
            rnd, "fp32-entries" if FP32 else "fp64", os.path.basename(path), sorted(k1)[len(k1) // 2], min(k1), nbytes / (sorted(k1)[len(k1) // 2] * 1e-6) / 1e12,
            nbytes / (sorted(k1)[len(k1) // 2] * 1e-6) / 8e12, sorted(fit)[len(fit) // 2], "same" if ll == ref else "DIFFER"), flush=True)
        g.close()
        ctx.close()

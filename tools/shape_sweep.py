#!/usr/bin/env python
"""GP-fit ms and EI evals/s at the BASELINE configs' shapes (one MI355X, candidates resident)."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from robo_amd import _lib
import bench

ctx = _lib.Context(0)
for N, D, M in ((1024, 8, 65536), (2048, 16, 65536), (4096, 16, 65536), (8192, 64, 65536)):
    X, y, theta, Xc = bench.synthetic(N, D, M, 0)
    g = _lib.DeviceGP(ctx, "matern52", N, D)
    g.set_data(X, y)
    c = float(y.mean())
    g.fit(theta, c)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); g.fit(theta, c); ts.append(time.perf_counter() - t0)
    cand = _lib.Candidates(ctx, Xc)
    eta = float(y.min())
    g.acq("ei", 0.0, eta, cand, want_values=False)
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        g.acq("ei", 0.0, eta, cand, want_values=False)
    ctx.synchronize()
    dt = (time.perf_counter() - t0) / 3
    fl = float(M) * N * N
    print("N=%d D=%d: fit %.3f ms; EI %d candidates in %.3f ms = %.2f M evals/s (%.1f TFLOP/s on the solve term)"
          % (N, D, min(ts) * 1e3, M, dt * 1e3, M / dt / 1e6, fl / dt / 1e12), flush=True)
    cand.close(); g.close()

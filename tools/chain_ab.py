"""Device-resident hyper-parameter chain (robo_gp_mcmc_run): time per ensemble half-step at BO-typical sizes, the fused
one-launch half-step (mcmc_block_step = 3: one- and two-block problems) against the launch-per-phase form (0), same session;
chains compared.       python tools/chain_ab.py [D] [walkers] [steps]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from robo_amd import _lib  # noqa: E402

D = int(sys.argv[1]) if len(sys.argv) > 1 else 16
K = int(sys.argv[2]) if len(sys.argv) > 2 else 52
STEPS = int(sys.argv[3]) if len(sys.argv) > 3 else 200
ctx = _lib.Context(0)
KNOB = os.environ.get("CHAIN_KNOB", "mcmc_block_step")          # which tuning key the two columns differ in
A_VAL, B_VAL = (int(v) for v in os.environ.get("CHAIN_VALUES", "3,0").split(","))
for N in (int(v) for v in os.environ.get("CHAIN_N", "100,127,150,200,254,255,300").split(",")):
    rs = np.random.RandomState(3)
    X = rs.rand(N, D)
    y = np.sinc(X * 10 - 5).sum(axis=1)
    P = D + 2
    p0 = np.concatenate([[0.5], np.full(D, np.log(0.25 * D)), [np.log(1e-3)]])[None, :] + 0.1 * rs.randn(K, P)
    uz, ua = rs.rand(STEPS, 2, K // 2), rs.rand(STEPS, 2, K // 2)
    pa = rs.randint(0, K // 2, size=(STEPS, 2, K // 2)).astype(np.int32)
    g = _lib.DeviceGP(ctx, "matern52", N, D)
    g.set_data(X, y)
    res = {}
    for mode in (A_VAL, B_VAL):
        ctx.set_tuning(KNOB, mode)
        g.mcmc_run(float(y.mean()), None, p0, None, 5, uz[:5], pa[:5], ua[:5])       # warm-up
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            out = g.mcmc_run(float(y.mean()), None, p0, None, STEPS, uz, pa, ua)
            ts.append(time.perf_counter() - t0)
        res[mode] = (min(ts), out)
    ctx.set_tuning(KNOB, None)
    (t3, o3), (t0_, o0) = res[A_VAL], res[B_VAL]
    same_pos = np.array_equal(o3[0], o0[0])
    print("N=%3d D=%d %d walkers x %d steps: %s=%d %.2f ms = %.1f us per half-step;  %s=%d %.2f ms = %.1f us;  "
          "final walkers %s, accepted %s, max |dlnp| %.2e" % (
              N, D, K, STEPS, KNOB, A_VAL, t3 * 1e3, t3 / (2 * STEPS + 2) * 1e6, KNOB, B_VAL, t0_ * 1e3,
              t0_ / (2 * STEPS + 2) * 1e6, "identical" if same_pos else "DIFFER",
              "same" if np.array_equal(o3[4], o0[4]) else "DIFFER", np.nanmax(np.abs(o3[1] - o0[1]))), flush=True)
    g.close()

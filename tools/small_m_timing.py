"""posterior + EI latency for small candidate batches at the headline GP (N=4096, D=16): the block-row substitution
(128- and 32-candidate steps) vs the explicit-inverse path (winv.hip); also a few smaller factors"""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from robo_amd import _lib
ctx = _lib.Context(0)


def best(f, reps=5):
    f()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); f(); ts.append((time.perf_counter() - t0) * 1e3)
    return min(ts)


for N, D in ((4096, 16), (2048, 16), (1000, 8), (500, 4)):
    X = np.random.RandomState(0).rand(N, D); y = np.sinc(X * 10 - 5).sum(axis=1); y = (y - y.mean()) / y.std()
    theta = np.concatenate([[0.0], np.full(D, np.log(0.25 * D)), [np.log(1e-3)]])
    g = _lib.DeviceGP(ctx, "matern52", N, D); g.set_data(X, y); g.fit(theta, 0.0)
    eta = float(y.min())
    t0 = time.perf_counter(); g.acq("ei", 0.0, eta, np.random.rand(500, D), want_values=False)
    print("N=%d: first 500-candidate call after the fit (allocates and builds W): %.3f ms" % (N, (time.perf_counter() - t0) * 1e3))
    # the W build alone, buffers already there: a refit, then cond_inf(L) (= triinv launches + two row-sum reductions + sync)
    wb = []
    for _ in range(4):
        g.fit(theta, 0.0)
        t0 = time.perf_counter(); cond = g.factor_cond()[0]; wb.append((time.perf_counter() - t0) * 1e3)
    t0 = time.perf_counter(); g.acq("ei", 0.0, eta, np.random.rand(500, D), want_values=False); t1 = (time.perf_counter() - t0) * 1e3
    g.fit(theta, 0.0)
    t0 = time.perf_counter(); g.acq("ei", 0.0, eta, np.random.rand(500, D), want_values=False); t2 = (time.perf_counter() - t0) * 1e3
    print("N=%d: W = L^-1 build + cond_inf(L) on a refitted factor: %.3f ms (cond_inf %.3g); 500-candidate call with W in place "
          "%.3f ms, right after a refit (build included) %.3f ms" % (N, min(wb), cond, t1, t2))
    for M in ((1, 8, 128, 500, 2048, 8192, 16384, 32768) if N == 4096 else (1, 500, 8192)):
        cand = _lib.Candidates(ctx, np.random.RandomState(1).rand(M, D))
        out = []
        for winv, small in ((0, 0), (0, 1000000), (1 << 30, 1000000)):
            ctx.set_tuning("winv_max", winv); ctx.set_tuning("winv_min_blocks", 1); ctx.set_tuning("trsm_small_max", small)
            out.append(best(lambda: g.acq("ei", 0.0, eta, cand, want_values=False)))
        kern = cand.solve_kernel()
        ctx.set_tuning("winv_gemv", 1)
        gemv_ms = best(lambda: g.acq("ei", 0.0, eta, cand, want_values=False)) if M <= 8 else float("nan")
        ctx.set_tuning("winv_gemv", 0)
        depths = []
        ctx.set_tuning("winv_rows", 0)
        for shift in (0, 1, 2):                  # the chunked form at its three unit depths
            ctx.set_tuning("winv_kc_shift", shift)
            depths.append(best(lambda: g.acq("ei", 0.0, eta, cand, want_values=False)))
        ctx.set_tuning("winv_kc_shift", None)
        forms = []
        for rows in (0, 1):                      # the two forms of the explicit-inverse product, forced
            ctx.set_tuning("winv_rows", rows)
            forms.append(best(lambda: g.acq("ei", 0.0, eta, cand, want_values=False)))
        for k in ("winv_max", "winv_min_blocks", "trsm_small_max", "winv_rows", "winv_gemv"):
            ctx.set_tuning(k, None)
        dflt = best(lambda: g.acq("ei", 0.0, eta, cand, want_values=False))
        print("N=%5d M=%6d: 128-cand step %.3f ms, 32-cand step %.3f ms, explicit inverse %.3f ms (%s; chunked units %.3f "
              "[depth full/half/quarter %.3f/%.3f/%.3f], whole-range rows %.3f, matrix-vector %.3f); default policy %.3f ms (%s)"
              % (N, M, out[0], out[1], out[2], kern, forms[0], depths[0], depths[1], depths[2], forms[1], gemv_ms, dflt,
                 cand.solve_kernel()))
        cand.close()
    g.close()

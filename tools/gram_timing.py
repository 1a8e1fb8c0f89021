"""K1 (gram kernel) time at the headline size for the kernel variants: one workgroup per tile vs persistent
workgroups (tuning key gram_persistent = workgroups per CU); HIP events 19 -> 21 around the kernel, 20 -> 21 the phase"""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from robo_amd import _lib
ctx = _lib.Context(0)
for N, D in ((4096, 16), (2048, 16), (8192, 64)):
    X = np.random.RandomState(0).rand(N, D); y = np.sinc(X * 10 - 5).sum(axis=1); y = (y - y.mean()) / y.std()
    theta = np.concatenate([[0.0], np.full(D, np.log(0.25 * D)), [np.log(1e-3)]])
    g = _lib.DeviceGP(ctx, "matern52", N, D); g.set_data(X, y)
    ctx.set_phase_events(True)
    nbytes = 8.0 * N * (N + 1) / 2 + 8.0 * N * D
    for label, mfma, wpc, half, occ in (("64x64 tiles (default)", 0, 0, 0, 0), ("64x64 tiles, 4 WG/CU", 0, 0, 0, 4), ("64x64 tiles, 5 WG/CU", 0, 0, 0, 5), ("64x64 tiles, 7 WG/CU", 0, 0, 0, 7),
                                        ("64x64 tiles, 8 WG/CU", 0, 0, 0, 8), ("32x64 tiles", 0, 0, 1, 0),
                                        ("64x64 tiles, mfma dots", 1, 0, 0, 0), ("64x64 tiles, persistent x4", 0, 4, 0, 0)):
        ctx.set_tuning("gram_occ", occ)
        ctx.set_tuning("gram_mfma", mfma)
        ctx.set_tuning("gram_half", half)
        ctx.set_tuning("gram_persistent", wpc)
        ks, ps = [], []
        for _ in range(6):
            g.fit(theta, 0.0)
            ks.append(ctx.elapsed_ms(19, 21)); ps.append(ctx.elapsed_ms(20, 21))
        k = min(ks)
        print("N=%d D=%d %-26s kernel %.1f us (%.2f TB/s, %.1f %% of 8 TB/s), phase %.1f us"
              % (N, D, label + ":", k * 1e3, nbytes / (k * 1e-3) / 1e12, 100 * nbytes / (k * 1e-3) / 8e12, min(ps) * 1e3))
    ctx.set_tuning("gram_persistent", None)
    ctx.set_tuning("gram_mfma", None)
    ctx.set_tuning("gram_half", None)
    ctx.set_tuning("gram_occ", None)
    ctx.set_phase_events(False)
    g.close()

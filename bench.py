#!/usr/bin/env python
"""Headline benchmark: EI evaluations/s (+ GP-fit ms) at N=4096, D=16 on 1..8 MI355X.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one candidate batch: from raw candidate
coordinates (already resident in HBM) through cross-gram, blocked triangular solve,
variance/mean, EI and the argmax, given a fitted GP.  With N > 1 ranks the
candidate axis is sharded (weak scaling: every rank evaluates its own --m candidates
against its own replica of the fitted GP); the only exchange is the all-gather of one
(max, index) pair per rank over RCCL.  GP-fit ms (gram + Cholesky + log-likelihood) is
reported next to it.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP64_MFMA_PEAK_TFLOPS = 78.6   # MI355X datasheet, dense fp64 matrix (= 32 FLOP/clk/SIMD * 1024 SIMD * 2.4 GHz)


def synthetic(N, D, M, rank):
    """SURVEY.md 8(d) synthetic inputs."""
    X = np.random.RandomState(0).rand(N, D)
    y = np.sinc(X * 10 - 5).sum(axis=1)
    y = (y - y.mean()) / y.std()
    theta = np.concatenate([[0.0], np.full(D, np.log(0.25 * D)), [np.log(1e-3)]])
    Xc = np.random.RandomState(1 + rank).rand(M, D)
    return X, y, theta, Xc


def cpu_baseline(N, D, theta, X, y, budget_s=20.0):
    """The reference CPU path restated by the oracle: gp.predict with the FULL covariance in
    batches of the reference's own 500 candidates (robo/maximizers/random_sampling.py:9,
    robo/models/gaussian_process.py:280-286) + EI, timed on a bounded sample."""
    from oracle import gp_oracle as O
    try:
        from threadpoolctl import threadpool_info
        cores = max([p.get("num_threads", 1) for p in threadpool_info()] + [1])
    except Exception:
        cores = os.cpu_count() or 1
    t0 = time.perf_counter()
    gp = O.OracleGP("matern52", theta, lower=np.zeros(D), upper=np.ones(D))
    gp.train(X, y)
    fit_s = time.perf_counter() - t0
    eta = y.min()
    rs = np.random.RandomState(99)
    done, t_pred = 0, 0.0
    gp.predict(rs.rand(500, D))   # warm-up
    while t_pred < budget_s and done < 20000:
        Xb = rs.rand(500, D)
        t0 = time.perf_counter()
        mu, var = gp.predict(Xb)          # full covariance then np.diag, like the reference
        O.ei(mu, var, eta)
        t_pred += time.perf_counter() - t0
        done += 500
    return {"value": done / t_pred, "unit": "EI evals/s", "cores": int(cores), "kind": "port",
            "sample": "%d candidates in batches of 500 (reference call sequence: full MxM covariance, np.diag), "
                      "N=%d D=%d; oracle fit %.0f ms" % (done, N, D, fit_s * 1e3),
            "gp_fit_ms": fit_s * 1e3}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--n", type=int, default=4096)
    ap.add_argument("--d", type=int, default=16)
    ap.add_argument("--m", type=int, default=65536, help="candidates per GPU")
    ap.add_argument("--acq", default="ei")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--lib", default=None, help="alternative build of librobo_hip.so (A/B runs of kernel variants)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1 or os.environ.get("ROBO_BENCH_FORCE_DIST") == "1":   # the latter: exercise the RCCL path on one rank
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", rank=rank, world_size=world)

    from robo_amd import _lib, sharding
    if args.lib:
        _lib.use_library(os.path.abspath(args.lib))

    ctx = _lib.Context(local_rank if world > 1 else int(os.environ.get("ROBO_DEVICE", "0")))
    N, D, M = args.n, args.d, args.m
    X, y, theta, Xc = synthetic(N, D, M, rank)
    mean_c = float(np.mean(y))
    eta = float(y.min())

    gp = _lib.DeviceGP(ctx, "matern52", N, D)
    gp.set_data(X, y)
    cand = _lib.Candidates(ctx, Xc)          # candidates resident in HBM before the timed region

    # ---- GP fit (replicated on every rank) ------------------------------------------------
    fit_ms, fit_chol_ms = [], []
    for i in range(3):
        t0 = time.perf_counter()
        gp.fit(theta, mean_c)
        fit_ms.append((time.perf_counter() - t0) * 1e3)
        fit_chol_ms.append((ctx.elapsed_ms(20, 21), ctx.elapsed_ms(21, 22), ctx.elapsed_ms(22, 23)))
    gram_ms, chol_ms, ll_ms = fit_chol_ms[int(np.argmin(fit_ms))]
    # the MCMC inner loop evaluates half an ensemble of thetas at once (n_hypers = 3 (D + 2) made
    # even = 54 at D = 16 -> 27 per half-step; robo/fmin/bayesian_optimization.py:85-87):
    # robo_gp_loglik_batch runs them through ONE sequence of launches
    S_half = max(1, (3 * (D + 2) + (3 * (D + 2)) % 2) // 2)
    thetas = theta[None, :] + 0.1 * np.random.RandomState(7).randn(S_half, theta.size)
    gp.loglik_batch(thetas, mean_c)
    t0 = time.perf_counter()
    gp.loglik_batch(thetas, mean_c)
    batch_ms = (time.perf_counter() - t0) * 1e3
    # analytic likelihood gradient (fit + W^T + A + reductions)
    gp.grad_loglik(theta, mean_c)
    t0 = time.perf_counter()
    gp.grad_loglik(theta, mean_c)
    grad_ms = (time.perf_counter() - t0) * 1e3
    grad_dev_ms = ctx.elapsed_ms(28, 29)
    gp.fit(theta, mean_c)          # the batch call leaves the GP unfitted

    def barrier():
        ctx.synchronize()
        if dist is not None:
            dist.barrier()
            import torch
            torch.cuda.synchronize()

    def step():
        _, mx, am, _ = gp.acq(args.acq, 0.0, eta, cand, want_values=False)
        return sharding.allgather_argmax(mx, am + rank * M)

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    trsm_ms = cross_ms = 0.0
    for _ in range(args.steps):
        best = step()
        cross_ms += ctx.elapsed_ms(24, 25)
        trsm_ms += ctx.elapsed_ms(25, 26)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        nb = (N + 127) // 128     # trsm_step_kernel launches per step (block rows holding training points)
        ms_per_step = elapsed / args.steps * 1e3
        value = world * M * args.steps / elapsed
        # dominant kernel: trsm_step_gen_kernel (nb launches per step; cross-gram tile generated in registers).  Algorithmic flops of one
        # step's launches: M * N^2 (the lower-triangular solve term of flops_ei);
        # per launch = M N^2 / nb; avg launch duration = trsm time / nb (HIP events on the
        # library's stream, slots 25->26).
        trsm_avg_launch_ms = trsm_ms / args.steps / nb
        achieved = (float(M) * N * N / nb) / (trsm_avg_launch_ms * 1e-3) / 1e12
        try:
            mb = ctx.microbench_mfma_f64_detail(4000)
            mfma_ceiling = mb["tflops"]
        except Exception:
            mb, mfma_ceiling = None, None
        # shader clock under an operand-streaming fp64 MFMA load (GEMM in the step kernel's shape): the
        # datasheet peak assumes 2.4 GHz, the chip holds 2.0-2.2 GHz on this kind of kernel
        try:
            g_tf, g_mhz = ctx.microbench_gemm_f64(0, 512, 2048, 3)
            clock_peak = FP64_MFMA_PEAK_TFLOPS * g_mhz / 2400.0
            gemm_mb = {"lds_core_tflops": g_tf, "shader_mhz_under_load": g_mhz,
                       "peak_at_that_clock_tflops": clock_peak, "frac_of_peak_at_that_clock": achieved / clock_peak}
        except Exception:
            gemm_mb = None
        traffic = None
        try:   # PMC-measured HBM bytes per launch for this exact workload (collected by tools/gpu_pmc.sh)
            tj = json.load(open(os.path.join(ROOT, "profiles", "trsm_traffic.json")))
            if tj["workload"] == {"n_train": N, "dim": D, "candidates_per_gpu": M}:
                traffic = tj["bytes_per_launch"]
        except Exception:
            pass
        out = {
            "metric": "EI evals/sec + GP-fit ms at N=4096,D=16; 1/2/4/8 MI355X vs host CPU",
            "value": value, "unit": "EI evals/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "GP Matern-5/2 ARD N=%d D=%d, %d uniform candidates per GPU, %s xi=0, fp64, "
                                   "candidate shard per GPU (BASELINE headline)" % (N, D, M, args.acq.upper()),
                       "n_train": N, "dim": D, "candidates_per_gpu": M, "acquisition": args.acq,
                       "parallelism": "candidate-shard x%d, replicated fit" % world},
            "gp_fit_ms": float(np.min(fit_ms)),
            "gp_fit_phases_ms": {"gram": gram_ms, "cholesky": chol_ms, "loglik": ll_ms},
            # north_star's two side figures: K-assembly against the HBM roofline (lower
            # triangle written once + X read once) and the factorisation against the fp64 MFMA peak
            "k_assembly": {"bytes": 8.0 * N * (N + 1) / 2 + 8.0 * N * D, "ms": gram_ms,
                           "GB_per_s": (8.0 * N * (N + 1) / 2 + 8.0 * N * D) / (gram_ms * 1e-3) / 1e9,
                           "frac_of_8TBps": (8.0 * N * (N + 1) / 2 + 8.0 * N * D) / (gram_ms * 1e-3) / 8.0e12},
            "cholesky": {"flops": N ** 3 / 3.0, "ms": chol_ms, "TFLOP_per_s": N ** 3 / 3.0 / (chol_ms * 1e-3) / 1e12,
                         "frac_of_mfma_peak": N ** 3 / 3.0 / (chol_ms * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS,
                         "note": "latency-bound by the 128 sequential pivots of each diagonal block; the trailing "
                                 "MFMA updates alone run at ~27 TFLOP/s (profiles/*_kernel_stats.csv)"},
            "gp_fit_batched": {"thetas": S_half, "ms_total": batch_ms, "ms_per_theta": batch_ms / S_half},
            "gp_grad_loglik_ms": {"total_incl_fit": grad_ms, "after_factorisation": grad_dev_ms},
            "ei_eval_phases_ms_per_step": {"cross_gram": cross_ms / args.steps, "trsm": trsm_ms / args.steps},
            "argmax": list(best),
            "roofline": {"bound": "mfma", "kernel": "trsm_step_gen_kernel", "achieved": achieved,
                         "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / FP64_MFMA_PEAK_TFLOPS,
                         "traffic": traffic, "traffic_unit": "bytes per launch (PMC, profiles/trsm_traffic.json)",
                         "algorithmic_flops_per_launch": float(M) * N * N / nb, "launches_per_step": nb, "avg_launch_ms": trsm_avg_launch_ms,
                         "mfma_f64_microbench_tflops": mfma_ceiling, "mfma_f64_microbench": mb,
                         "gemm_f64_microbench": gemm_mb},
            "device": ctx.name,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(N, D, theta, X, y)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""Headline benchmark: EI evaluations/s (+ GP-fit ms) at N=4096, D=16 on 1..8 MI355X.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W          # one process per GPU (the driver's form)
    python bench.py --gpus N                      # the same without a launcher: starts its N rank processes itself
    python bench.py --gpus N --launcher inproc    # ONE process drives all N GPUs (robo_amd/csrc/multi.hip): the form
                                                  # robo_amd.fmin.*(n_gpus=N) runs; same line shape

A "step" is one pass of the hot path over one candidate batch: from raw candidate
coordinates (already resident in HBM) through cross-gram, blocked triangular solve,
variance/mean, EI and the argmax, given a fitted GP.  With N > 1 ranks the
candidate axis is sharded (weak scaling: every rank evaluates its own --m candidates
against its own replica of the fitted GP); the only exchange is the all-gather of one
(max, index) pair per rank over RCCL.  GP-fit ms (gram + Cholesky + log-likelihood) is
reported next to it.  Prints ONE JSON line on rank 0.

``--config`` selects one of BASELINE.json's other configurations (same JSON shape, their own workload and
roofline; the default, and what the driver runs, is the headline):
    c2   GP Matern-5/2 N=1024 D=8, 65 536 candidates, EI                          (candidate shard)
    c3   50 hyper-parameter samples, N=2048 D=16, marginal LogEI over 65 536      (SAMPLE shard, 13/13/12/12 at 4)
    c4   Fabolas kernel N=4096 D=10+1, information gain per unit cost, 8192/GPU   (candidate shard)
    c5   N=8192 D=64, LCB, 2^17 Sobol candidates per GPU, fp32 K-build            (candidate shard;
         --m 1048576: the whole 2^20-candidate set on one GPU, 70 GB solve workspace in one pass)
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
METRIC = "EI evals/sec + GP-fit ms at N=4096,D=16; 1/2/4/8 MI355X vs host CPU"


def default_theta(D, extra=0):
    """log amp 0, log l^2 = log(0.25 D), log sigma^2 = log 1e-3"""
    return np.concatenate([[0.0], np.full(D, np.log(0.25 * D)), np.zeros(extra), [np.log(1e-3)]])


def synthetic(N, D, M, rank):
    """SURVEY.md 8(d) synthetic inputs."""
    X = np.random.RandomState(0).rand(N, D)
    y = np.sinc(X * 10 - 5).sum(axis=1)
    y = (y - y.mean()) / y.std()
    Xc = np.random.RandomState(1 + rank).rand(M, D)
    return X, y, default_theta(D), Xc


def flops_ei(N, D):
    """algorithmic flops of one acquisition evaluation"""
    return float(N) * N + N * (3.0 * D + 20.0)


def cpu_baseline(N, D, theta, X, y, budget_s=12.0):
    """The reference CPU path as a RESTATEMENT (george is not installable here): the oracle makes
    the reference's call sequence -- gp.predict with the FULL covariance in batches of the reference's own 500
    candidates (robo/maximizers/random_sampling.py:9, robo/models/gaussian_process.py:280-286) + EI -- timed on a
    bounded sample; next to it the "fair" diag-only variant (no M x M covariance), so that the GPU/CPU ratio is
    not inflated by the reference's wasted O(M^2) work."""
    from oracle import gp_oracle as O
    try:
        from threadpoolctl import threadpool_info
        blas_threads = max([p.get("num_threads", 1) for p in threadpool_info()] + [1])
    except Exception:
        blas_threads = os.cpu_count() or 1
    t0 = time.perf_counter()
    gp = O.OracleGP("matern52", theta, lower=np.zeros(D), upper=np.ones(D))
    gp.train(X, y)
    fit_s = time.perf_counter() - t0
    eta = y.min()
    rs = np.random.RandomState(99)
    gp.predict(rs.rand(500, D))   # warm-up
    done, t_pred = 0, 0.0
    while t_pred < budget_s and done < 20000:
        Xb = rs.rand(500, D)
        t0 = time.perf_counter()
        mu, var = gp.predict(Xb)          # full covariance then np.diag, like the reference
        O.ei(mu, var, eta)
        t_pred += time.perf_counter() - t0
        done += 500
    fair_done, t_fair = 0, 0.0
    while t_fair < budget_s / 2 and fair_done < 65536:
        Xb = rs.rand(4096, D)
        t0 = time.perf_counter()
        mu, var = gp.predict(Xb, diag_only=True)
        O.ei(mu, var, eta)
        t_fair += time.perf_counter() - t0
        fair_done += 4096
    return {"value": done / t_pred, "unit": "EI evals/s", "cores": int(blas_threads), "kind": "port",
            "os_cpu_count": os.cpu_count(),
            "note": "restatement of the george call sequence on NumPy/SciPy (george itself is not installable: "
                    "BASELINE.md 3.5); cores = BLAS threads in use",
            "sample": "%d candidates in batches of 500 (reference call sequence: full MxM covariance, np.diag), "
                      "N=%d D=%d; oracle fit %.0f ms" % (done, N, D, fit_s * 1e3),
            "fair_diag_only": {"value": fair_done / t_fair, "unit": "EI evals/s",
                               "sample": "%d candidates in batches of 4096, diagonal variance only" % fair_done},
            "gp_fit_ms": fit_s * 1e3}


def _blas_threads():
    try:
        from threadpoolctl import threadpool_info
        return int(max([p.get("num_threads", 1) for p in threadpool_info()] + [1]))
    except Exception:
        return os.cpu_count() or 1


def cpu_baseline_c3(N, D, M, thetas, X, y, budget_s=14.0):
    """config 3 on the host cores, the reference's call sequence restated (george is not installable): per hyper-
    parameter sample one gp.compute (gaussian_process_mcmc.py:149-164 -> gaussian_process.py:119) and, per 500-candidate
    RandomSampling batch, gp.predict with the full covariance + LogEI's per-point Python loop (log_ei.py:79-120);
    MarginalizationGPMCMC averages the S vectors (marginalization.py:115-121).  Bounded sample: a few samples' fits and a
    few batches each; the rate for the config's shape (one fit per sample per 65 536 candidates) follows from the two."""
    from oracle import gp_oracle as O
    eta = float(y.min())
    rs = np.random.RandomState(98)
    t_fit, t_eval, n_eval, n_fit = 0.0, 0.0, 0, 0
    while t_fit + t_eval < budget_s and n_fit < len(thetas):
        t0 = time.perf_counter()
        gp = O.OracleGP("matern52", thetas[n_fit], lower=np.zeros(D), upper=np.ones(D))
        gp.train(X, y)
        t_fit += time.perf_counter() - t0
        n_fit += 1
        for _ in range(2):
            Xb = rs.rand(500, D)
            t0 = time.perf_counter()
            mu, var = gp.predict(Xb)                     # full covariance, then np.diag, like the reference
            O.log_ei(mu, var, eta)                       # the reference's per-point loop
            t_eval += time.perf_counter() - t0
            n_eval += 500
    per_fit, per_eval = t_fit / n_fit, t_eval / n_eval
    return {"value": M / (per_fit + M * per_eval), "unit": "LogEI sample-evals/s", "cores": _blas_threads(),
            "kind": "port", "os_cpu_count": os.cpu_count(),
            "sample": "%d samples' fits (%.0f ms each, N=%d) and %d candidates in batches of 500 (full MxM covariance + "
                      "LogEI's Python loop, %.1f us per candidate); value = the rate of one sample's %d candidates "
                      "incl. its fit" % (n_fit, per_fit * 1e3, N, n_eval, per_eval * 1e6, M),
            "note": "restatement of the george call sequence on NumPy/SciPy (BASELINE.md 3.5)"}


def cpu_baseline_c4(N, D, theta, X, y, Xcost, cost, Xc, Xc_cost, zb, ep_arrays, W, sn2, budget_s=14.0):
    """config 4 on the host cores: the reference's per-candidate loop (information_gain.py:112 over
    information_gain_per_unit_cost.py:91-104) restated -- per candidate x one model.predict(x) and one
    predict_variance(rep, x) = a full (Nb+1) x (Nb+1) posterior covariance through K^-1 (gaussian_process.py:243-248),
    the innovation algebra (_dh_fun) and the cost model's predict."""
    from oracle import gp_oracle as O
    from oracle import ig_oracle as IG
    t0 = time.perf_counter()
    gp = O.OracleGP("fabolas", theta, normalize_input=False)
    gp.train(X, y)
    gc = O.OracleGP("fabolas", theta, normalize_input=False)
    gc.train(Xcost, cost)
    fit_s = time.perf_counter() - t0
    logP, lmb, dMu, dSig, dMM = ep_arrays
    done, t = 0, 0.0
    vals = []
    while t < budget_s and done < Xc.shape[0]:
        x, xc = Xc[done:done + 1], Xc_cost[done:done + 1]
        t0 = time.perf_counter()
        v = gp.predict(x)[1][0]
        cov = gp.predict(np.concatenate((zb, x)), full_cov=True)[1]
        dh = IG.dh_fun(v, cov[-1, :-1][:, None], sn2, logP, lmb, dMu, dSig, dMM, W)
        vals.append(dh / np.exp(gc.predict(xc)[0][0]))
        t += time.perf_counter() - t0
        done += 1
    return {"value": done / t, "unit": "information gains/s", "cores": _blas_threads(), "kind": "port",
            "os_cpu_count": os.cpu_count(), "gp_fit_ms": fit_s * 1e3 / 2,
            "sample": "%d candidates, one at a time like the reference (predict + predict_variance over Nb=%d "
                      "representers + dH + cost predict), N=%d" % (done, zb.shape[0], N),
            "note": "restatement of the george call sequence on NumPy/SciPy (BASELINE.md 3.5)",
            "_values": vals}


def cpu_baseline_c5(N, D, theta, X, y, budget_s=12.0):
    """config 5 on the host cores: fp64 throughout (the reference has no mixed precision): gp.compute at N=8192, then
    gp.predict (full covariance) + LCB on 500-candidate batches; diag-only variant next to it."""
    from oracle import gp_oracle as O
    t0 = time.perf_counter()
    gp = O.OracleGP("matern52", theta, lower=np.zeros(D), upper=np.ones(D))
    gp.train(X, y)
    fit_s = time.perf_counter() - t0
    rs = np.random.RandomState(97)
    done, t = 0, 0.0
    while t < budget_s and done < 5000:
        Xb = rs.rand(500, D)
        t0 = time.perf_counter()
        mu, var = gp.predict(Xb)
        O.lcb(mu, var)
        t += time.perf_counter() - t0
        done += 500
    fd, tf = 0, 0.0
    while tf < budget_s / 2 and fd < 16384:
        Xb = rs.rand(2048, D)
        t0 = time.perf_counter()
        mu, var = gp.predict(Xb, diag_only=True)
        O.lcb(mu, var)
        tf += time.perf_counter() - t0
        fd += 2048
    return {"value": done / t, "unit": "LCB evals/s", "cores": _blas_threads(), "kind": "port",
            "os_cpu_count": os.cpu_count(), "gp_fit_ms": fit_s * 1e3,
            "sample": "%d candidates in batches of 500 (reference call sequence: full MxM covariance, np.diag), N=%d "
                      "D=%d, all fp64; oracle fit %.0f ms" % (done, N, D, fit_s * 1e3),
            "fair_diag_only": {"value": fd / tf, "unit": "LCB evals/s",
                               "sample": "%d candidates in batches of 2048, diagonal variance only" % fd},
            "note": "restatement of the george call sequence on NumPy/SciPy (BASELINE.md 3.5)"}


class TorchExchange(object):
    """bench.py's safety net only (see Dist.make_comm): the two exchanges over torch.distributed, host-mediated"""

    def __init__(self, dist, rank, world):
        self.dist, self.rank, self.world = dist, rank, world

    def allgather(self, values):
        import torch
        mine = torch.tensor(np.asarray(values, dtype=np.float64).reshape(-1), dtype=torch.float64, device="cuda")
        out = [torch.empty_like(mine) for _ in range(self.world)]
        self.dist.all_gather(out, mine)
        return np.stack([t.cpu().numpy() for t in out])

    def acq_sharded(self, gp, kind, par, eta, cand, global_offset, want_values=False):
        from robo_amd import sharding
        vals, mx, am, fl = gp.acq(kind, par, eta, cand, want_values)
        rows = self.allgather([mx, float(am + global_offset), float(fl)])
        best = sharding.reduce_argmax((float(r[0]), int(r[1])) for r in rows)
        flags = 0
        for r in rows:
            flags |= int(r[2])
        owner = [i for i, r in enumerate(rows) if int(r[1]) == best[1]][0]
        return vals, best[0], best[1], owner, flags

    def acq_marginal_sharded(self, gps, s_total, kind, par, etas, cand, want_values=True):
        from robo_amd import _lib
        part, _, _, fl = _lib.acq_marginal(gps, kind, par, etas, cand, reduce="sum")
        rows = self.allgather(part)
        total = rows[0].copy()
        for r in rows[1:]:
            total += r
        total /= s_total
        j = int(np.argmax(total))
        return (total if want_values else None), float(total[j]), j, fl

    def close(self):
        pass


class Dist(object):
    """One process per GPU.  Three ways into a multi-rank run, all ending in the library's own communicator
    (robo_amd/csrc/comm.hip, RCCL all-gathers on the library's stream):

    * ``spawn``  -- ``python bench.py --gpus N`` without WORLD_SIZE: the parent started this process as one of N ranks
      (launch_ranks) and the 128-byte communicator id travels through a file in a private temporary directory.  No torch.
    * ``torch``  -- under ``python -m torch.distributed.run``: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the
      environment; torch.distributed (backend nccl = RCCL) carries the id, nothing else.
    * ``single`` -- one process, no communicator (ROBO_BENCH_FORCE_DIST=1: a one-rank communicator, the RCCL path alone).
    """

    def __init__(self):
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.dist = None
        self.out_fd = None
        self.comm = None
        self.exchange = None
        self.rdv_dir = os.environ.get("ROBO_BENCH_RENDEZVOUS") or None
        force = os.environ.get("ROBO_BENCH_FORCE_DIST") == "1"     # the RCCL path on one rank
        if self.rdv_dir:
            self.mode = "spawn"
        elif self.world > 1 or force:
            self.mode = "torch"
        else:
            self.mode = "single"
        if self.mode != "single":
            # RCCL prints a version banner on the C-level stdout; the contract is ONE JSON line there.  Keep the
            # real stdout aside for that line and send everything else libraries write to fd 1 to stderr.
            sys.stdout.flush()
            self.out_fd = os.dup(1)
            os.dup2(2, 1)
        if self.mode == "torch":
            import torch
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            torch.cuda.set_device(self.local_rank)
            dist.init_process_group(backend="nccl", rank=self.rank, world_size=self.world,
                                    device_id=torch.device("cuda", self.local_rank))
            self.dist = dist

    @property
    def multi(self):
        """a communicator takes part in every step (also the forced one-rank form)"""
        return self.mode != "single"

    def device_ordinal(self):
        return self.local_rank if self.multi and self.world > 1 else int(os.environ.get("ROBO_DEVICE", "0"))

    def _comm_id(self, _lib):
        """the 128-byte id: created on rank 0, to every rank out of band"""
        if self.mode == "spawn":
            path = os.path.join(self.rdv_dir, "comm_id.bin")
            if self.rank == 0:
                blob = _lib.Comm.create_id()
                with open(path + ".tmp", "wb") as f:
                    f.write(blob)
                os.rename(path + ".tmp", path)           # atomic: a reader never sees a partial id
                return blob
            deadline = time.time() + float(os.environ.get("ROBO_BENCH_COMM_TIMEOUT", "240"))
            while not os.path.exists(path):
                if time.time() > deadline:
                    raise RuntimeError("rank %d: no communicator id from rank 0 within the deadline" % self.rank)
                time.sleep(0.01)
            with open(path, "rb") as f:
                return f.read()
        box = [_lib.Comm.create_id() if self.rank == 0 else None]
        if self.world > 1:
            self.dist.broadcast_object_list(box, src=0)
        return box[0]

    def make_comm(self, _lib, ctx):
        """the library's own communicator (robo_amd/csrc/comm.hip: RCCL all-gathers on the library's stream) for the
        exchanges of the data path; the launcher (file / torch.distributed) only carries its 128-byte id"""
        if not self.multi:
            return None
        err = ""
        try:
            blob = self._comm_id(_lib)
            # ncclCommInitRank blocks until every rank has joined: run it on a helper thread with a deadline, so that a
            # rank that cannot join turns into an error (spawn) / the fallback below (torch) instead of a hung run
            import threading
            res = {}

            def _init():
                try:
                    res["comm"] = _lib.Comm(ctx, self.rank, self.world, blob)
                except Exception as e:   # noqa: BLE001
                    res["err"] = "%s: %s" % (type(e).__name__, e)

            th = threading.Thread(target=_init, daemon=True)
            th.start()
            th.join(float(os.environ.get("ROBO_BENCH_COMM_TIMEOUT", "240")))
            comm = res.get("comm")
            if comm is None:
                err = res.get("err", "communicator setup timed out")
        except Exception as e:           # noqa: BLE001 -- measured anyway, and said so in the JSON line
            comm, err = None, "%s: %s" % (type(e).__name__, e)
        if self.mode == "spawn":
            if comm is None:
                raise RuntimeError("rank %d: library communicator unavailable (%s)" % (self.rank, err))
        else:
            import torch
            ok = torch.tensor([1.0 if comm is not None else 0.0], device="cuda")
            self.dist.all_reduce(ok, op=self.dist.ReduceOp.MIN)
            if float(ok.item()) < 1.0:
                # bench-only safety net (NOT part of the product): if the in-library communicator cannot be set up on
                # this node, the same exchanges go through torch.distributed so that the scaling run still yields a
                # measurement
                if comm is not None:
                    comm.close()
                sys.stderr.write("bench: library communicator unavailable (%s); torch.distributed exchange\n" % err)
                self.exchange = "torch.distributed fallback (%s)" % (err or "another rank failed")
                self.comm = TorchExchange(self.dist, self.rank, self.world)
                return self.comm
        self.exchange = "librobo_hip (RCCL all-gather on the library's stream, device pointers); id via %s" % (
            "a file of the self-launcher" if self.mode == "spawn" else "torch.distributed")
        self.comm = comm
        from robo_amd import sharding
        sharding._comm = self.comm        # the module-level helpers (allgather_argmax ...) use this communicator too
        return self.comm

    def _lib_comm(self):
        c = self.comm
        return c if c is not None and not isinstance(c, TorchExchange) else None

    def ranks_block(self, ctx):
        """who took part, as the COMMUNICATOR saw it: world from robo_comm_info, every rank's (rank, device ordinal)
        all-gathered through it.  COLLECTIVE in a multi-rank run (call it on every rank, outside the timed region)."""
        c = self._lib_comm()
        if c is None:
            if isinstance(self.comm, TorchExchange):
                rows = self.comm.allgather([float(self.rank), float(ctx.device)])
                return {"ranks": [int(r[0]) for r in rows], "devices": [int(r[1]) for r in rows],
                        "comm_world": self.world, "launcher": self.mode}
            return {"ranks": [0], "devices": [int(ctx.device)], "comm_world": 1, "launcher": self.mode}
        rk, wd = c.info()
        rows = c.allgather([float(rk), float(ctx.device)])
        return {"ranks": [int(r[0]) for r in rows], "devices": [int(r[1]) for r in rows], "comm_world": int(wd),
                "launcher": self.mode}

    def barrier(self, ctx):
        """all ranks' device work finished: stream synchronisation + a collective on every rank.  With the library's
        communicator that collective is its own 8-byte all-gather on the library's stream (~30 us); torch's NCCL barrier
        (measured at several hundred us inside a 5-step timed region, r03zk) only when that communicator is absent."""
        ctx.synchronize()
        if not self.multi:
            return
        c = self._lib_comm()
        if c is not None:
            c.allgather(np.zeros(1))
            ctx.synchronize()
            return
        import torch
        self.dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(self, seconds):
        if not self.multi:
            return seconds
        c = self._lib_comm()
        if c is not None:
            return float(np.max(c.allgather(np.array([seconds]))))
        import torch
        t = torch.tensor([seconds], dtype=torch.float64, device="cuda")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def emit(self, obj):
        line = (json.dumps(obj) + "\n").encode()
        if self.out_fd is None:
            sys.stdout.write(line.decode())
            sys.stdout.flush()
        else:
            os.write(self.out_fd, line)

    def close(self):
        if self._lib_comm() is not None:
            from robo_amd import sharding
            sharding.close_comm()
        if self.dist is not None:
            self.dist.barrier()
            self.dist.destroy_process_group()


def launch_ranks(n, argv):
    """``python bench.py --gpus N`` without a launcher around it: start N rank processes of this script (one per GPU,
    RANK / LOCAL_RANK / WORLD_SIZE in their environment, a private directory for the communicator id), wait for all of
    them, pass rank 0's JSON line through.  Returns the exit status (non-zero if any rank failed)."""
    import shutil
    import subprocess
    import tempfile
    rdv = tempfile.mkdtemp(prefix="robo_bench_")
    procs = []
    try:
        for r in range(n):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), ROBO_BENCH_RENDEZVOUS=rdv)
            env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC only on this host driver (RCCL needs it)
            # rank 0 inherits stdout (the ONE JSON line); the other ranks' stdout goes to stderr
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                          stdout=None if r == 0 else sys.stderr))
        deadline = time.time() + float(os.environ.get("ROBO_BENCH_LAUNCH_TIMEOUT", "3000"))
        status = 0
        pending = list(procs)
        while pending:
            for p in list(pending):
                rc = p.poll()
                if rc is not None:
                    pending.remove(p)
                    if rc != 0:
                        status = status or rc
            if status != 0 or time.time() > deadline:
                # a rank died (or the run hung): the others would wait in a collective for ever -- stop exactly the
                # processes started here
                time.sleep(2.0)
                for p in pending:
                    if p.poll() is None:
                        p.terminate()
                for p in pending:
                    try:
                        p.wait(timeout=10)
                    except Exception:      # noqa: BLE001
                        p.kill()
                status = status or 124
                break
            time.sleep(0.05)
        return status
    finally:
        shutil.rmtree(rdv, ignore_errors=True)


def _hip_runtime():
    import ctypes
    for name in ("libamdhip64.so", "libamdhip64.so.7", "libamdhip64.so.6", "/opt/rocm/lib/libamdhip64.so"):
        try:
            return ctypes.CDLL(name)
        except OSError:
            continue
    return None


def _pin_host(arr):
    """page-lock a NumPy array in place (hipHostRegister); False when the runtime is not loadable or declines"""
    import ctypes
    hip = _hip_runtime()
    if hip is None or not hasattr(hip, "hipHostRegister"):
        return False
    hip.hipHostRegister.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint]
    return hip.hipHostRegister(ctypes.c_void_p(arr.ctypes.data), arr.nbytes, 0) == 0


def _unpin_host(arr):
    import ctypes
    hip = _hip_runtime()
    if hip is not None and hasattr(hip, "hipHostUnregister"):
        hip.hipHostUnregister.argtypes = [ctypes.c_void_p]
        hip.hipHostUnregister(ctypes.c_void_p(arr.ctypes.data))


def timed_steps(D_, ctx, step, steps, warmup):
    for _ in range(warmup):
        step()
    D_.barrier(ctx)
    t0 = time.perf_counter()
    last = None
    trsm_ms = 0.0
    for _ in range(steps):
        last = step()
        trsm_ms += ctx.elapsed_ms(25, 26)
    D_.barrier(ctx)
    return D_.max_over_ranks(time.perf_counter() - t0), last, trsm_ms / steps


def config_traffic(config, kernel):
    """PMC-measured HBM bytes per launch of `kernel` for one of the other configurations (profiles/trsm_traffic_<config>.json,
    written by tools/gpu_r03.sh pmcconf), or None"""
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "trsm_traffic_%s.json" % config)))
        return tj["bytes_per_launch"] if tj["kernel"] in kernel or kernel in tj["kernel"] else None
    except Exception:
        return None


def hyper_inference(with_cpu, N=200, first_iteration=True, cpu_budget_s=2.0):
    """SURVEY 8(a2, a5) / 8(f) rank 1 -- the step BEFORE the posterior, where the fit multiplies: the hyper-parameter
    inference of ONE Bayesian-optimisation iteration with the reference's defaults (robo/fmin/bayesian_optimization.py:
    75-100: 2 * ARD Matern-5/2, DefaultPrior, n_hypers = 3 len(kernel) made even, burn-in 100 + chain 200 ensemble steps,
    i.e. 15 700 likelihoods of the first iteration at D = 16, 10 450 of every later one) at a BO-typical N = 200:
    robo_amd's GaussianProcessMCMC.train (the whole chain on the device, robo_gp_mcmc_run), wall clock, against the
    oracle's loglikelihood (the reference's per-walker george fit, restated) on a sample of walkers."""
    from robo_amd.kernels import Matern52Kernel
    from robo_amd.priors import DefaultPrior
    from robo_amd.models.gaussian_process_mcmc import GaussianProcessMCMC
    D = 16
    rs = np.random.RandomState(5)
    X = rs.rand(N, D)
    y = np.sinc(X * 10 - 5).sum(axis=1)
    kernel = 2 * Matern52Kernel(np.ones([D]), ndim=D)
    prior = DefaultPrior(len(kernel) + 1, rng=np.random.RandomState(6))
    nh = 3 * len(kernel)
    nh += nh % 2
    model = GaussianProcessMCMC(kernel, prior=prior, n_hypers=nh, chain_length=200, burnin_steps=100,
                                normalize_input=True, normalize_output=False, rng=np.random.RandomState(7),
                                lower=np.zeros(D), upper=np.ones(D))
    first = None
    if first_iteration:
        t0 = time.perf_counter()
        model.train(X, y)
        first = time.perf_counter() - t0
    else:
        # large N: the burn-in (100 more ensemble steps of the same cost per step) is not run; the chain starts from the
        # prior's samples like a burn-in would
        model.p0 = prior.sample_from_prior(nh)
        model.burned = True
    t0 = time.perf_counter()
    model.train(X, y)                        # burned: chain only -- what every later BO iteration pays
    later = time.perf_counter() - t0
    n_first, n_later = nh * (100 + 200 + 2), nh * (200 + 1)
    out = {"n_train": N, "dim": D, "walkers": nh, "what": "GaussianProcessMCMC.train with the reference's defaults "
           "(burn-in 100 + chain 200 ensemble steps; later iterations: chain only), incl. the %d per-sample fits" % nh,
           "first_iteration_ms": None if first is None else first * 1e3, "later_iteration_ms": later * 1e3,
           "likelihoods_first": n_first, "likelihoods_later": n_later,
           "likelihoods_per_s": n_later / later,
           # SURVEY 8(d)'s GP-fit work per likelihood (N^3 / 3 + K assembly) against the fp64 MFMA peak
           "frac_of_fp64_mfma_peak": n_later * (N ** 3 / 3.0 + N * (N + 1) / 2.0 * (3 * D + 16) + 2.0 * N * N)
           / later / 1e12 / FP64_MFMA_PEAK_TFLOPS}
    if with_cpu:
        from oracle import gp_oracle as O
        hyp = np.asarray(model.hypers)
        ogp = O.OracleGP("matern52", hyp[0], lower=np.zeros(D), upper=np.ones(D))
        ogp.train(X, y)
        ogp.loglikelihood(hyp[0])
        t0 = time.perf_counter()
        cnt = 0
        while time.perf_counter() - t0 < cpu_budget_s:
            ogp.loglikelihood(hyp[cnt % len(hyp)])
            cnt += 1
        per = (time.perf_counter() - t0) / cnt
        out["cpu_port"] = {"ms_per_likelihood": per * 1e3, "likelihoods_timed": cnt,
                           "later_iteration_ms_at_that_rate": per * n_later * 1e3,
                           "note": "oracle restatement of gaussian_process_mcmc.py:168-202 (K build + cho_factor + "
                                   "log-likelihood per walker), single process"}
    return out


def bo_iteration(N=4096, D=16):
    """One iteration of the reference's loop at the headline size with ITS defaults (robo/solver/bayesian_optimization.py:
    236-245: model.train -> acquisition_func.update -> maximize_func.maximize with RandomSampling's 500 candidates,
    robo/maximizers/random_sampling.py:9), through the product classes, wall clock: what a user of robo.fmin pays per
    iteration beside the objective function.  do_optimize=False: hyper-parameter inference is timed separately
    (hyper_inference)."""
    from robo_amd.kernels import Matern52Kernel
    from robo_amd.models import GaussianProcess
    from robo_amd.acquisition_functions import EI
    from robo_amd.maximizers import RandomSampling
    rs = np.random.RandomState(0)
    X = rs.rand(N, D)
    y = np.sinc(X * 10 - 5).sum(axis=1)
    y = (y - y.mean()) / y.std()
    theta = default_theta(D)
    kernel = Matern52Kernel(np.exp(theta[1:-1]), ndim=D, log_amp=theta[0])
    model = GaussianProcess(kernel, noise=1e-3, lower=np.zeros(D), upper=np.ones(D), rng=np.random.RandomState(1))
    acq = EI(model)
    maxi = RandomSampling(acq, np.zeros(D), np.ones(D), rng=np.random.RandomState(2))
    ts = {"train": [], "update": [], "maximize": [], "total": []}
    for it in range(6):
        np.random.seed(it)
        t0 = time.perf_counter()
        model.train(X, y, do_optimize=False)
        t1 = time.perf_counter()
        acq.update(model)
        t2 = time.perf_counter()
        maxi.maximize()
        t3 = time.perf_counter()
        if it:                                   # the first iteration allocates
            for k_, v_ in (("train", t1 - t0), ("update", t2 - t1), ("maximize", t3 - t2), ("total", t3 - t0)):
                ts[k_].append(v_ * 1e3)
    return {"n_train": N, "dim": D, "candidates": 500, "what": "GaussianProcess.train(do_optimize=False) + EI.update + "
            "RandomSampling.maximize (the reference's recipe and default of 500 candidates), median of 5 iterations after the first",
            "ms": {k_: float(np.median(v_)) for k_, v_ in ts.items()}}


def bo_iteration_gp_mcmc(N=2048, D=16):
    """One iteration of robo.fmin.bayesian_optimization's DEFAULT configuration (model_type="gp_mcmc", acquisition
    "log_ei", maximizer "random"; robo/fmin/bayesian_optimization.py:27-30,85-139) at BASELINE config 3's model size, end to
    end through the product classes: GaussianProcessMCMC.train (chain of 200 ensemble steps with 54 walkers -- a later
    iteration, burn-in done -- and the 54 per-sample fits) + MarginalizationGPMCMC.update + RandomSampling.maximize (500
    candidates x 54 samples), wall clock."""
    from robo_amd.kernels import Matern52Kernel
    from robo_amd.priors import DefaultPrior
    from robo_amd.models import GaussianProcessMCMC
    from robo_amd.acquisition_functions import LogEI, MarginalizationGPMCMC
    from robo_amd.maximizers import RandomSampling
    rs = np.random.RandomState(5)
    X = rs.rand(N, D)
    y = np.sinc(X * 10 - 5).sum(axis=1)
    kernel = 2 * Matern52Kernel(np.ones([D]), ndim=D)
    prior = DefaultPrior(len(kernel) + 1, rng=np.random.RandomState(6))
    nh = 3 * len(kernel)
    nh += nh % 2
    model = GaussianProcessMCMC(kernel, prior=prior, n_hypers=nh, chain_length=200, burnin_steps=100,
                                normalize_input=True, normalize_output=False, rng=np.random.RandomState(7),
                                lower=np.zeros(D), upper=np.ones(D))
    model.p0 = prior.sample_from_prior(nh)
    model.burned = True
    model.chain_length = 3
    model.train(X, y)                            # allocations, first-use builds
    model.chain_length = 200
    acq = MarginalizationGPMCMC(LogEI(model))
    maxi = RandomSampling(acq, np.zeros(D), np.ones(D), rng=np.random.RandomState(2))
    np.random.seed(0)
    t0 = time.perf_counter()
    model.train(X, y)
    t1 = time.perf_counter()
    acq.update(model)
    t2 = time.perf_counter()
    maxi.maximize()
    t3 = time.perf_counter()
    return {"n_train": N, "dim": D, "walkers": nh, "samples": nh, "candidates": 500, "chain_steps": 200,
            "what": "GaussianProcessMCMC.train (200 ensemble steps, 54 walkers, + 54 per-sample fits) + "
                    "MarginalizationGPMCMC(LogEI).update + RandomSampling.maximize: one later iteration of "
                    "robo.fmin.bayesian_optimization's default configuration, one run",
            "ms": {"train": (t1 - t0) * 1e3, "update": (t2 - t1) * 1e3, "maximize": (t3 - t2) * 1e3,
                   "total": (t3 - t0) * 1e3}}


def roofline_trsm(ctx, N, M_rows, trsm_ms_per_step, passes=1, kernel="trsm_step_gen_kernel", traffic=None):
    """the dominant kernel of every configuration is the block-row solve: algorithmic flops per launch =
    rows N^2 / nb (SURVEY.md 8d's triangular-solve term), duration from HIP events on the library's stream
    (slots 25 -> 26) averaged over the nb launches of a pass"""
    nb = (N + 127) // 128
    if kernel.startswith("winv_"):
        # the explicit-inverse path is ONE triangular product per pass; the event pair brackets cross-gram + product +
        # chunk reduction, priced against the product's algorithmic flops (rows N^2)
        nb = 1
    avg_launch_ms = trsm_ms_per_step / (nb * passes)
    if not avg_launch_ms > 0.0:          # no event pair (the interpreter build's clock, or a path that records none)
        return {"bound": "mfma", "kernel": kernel, "achieved": None, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": None, "traffic": traffic, "algorithmic_flops_per_launch": float(M_rows) * N * N / nb,
                "launches_per_step": nb * passes, "avg_launch_ms": None}
    achieved = (float(M_rows) * N * N / nb) / (avg_launch_ms * 1e-3) / 1e12
    return {"bound": "mfma", "kernel": kernel, "achieved": achieved, "peak": FP64_MFMA_PEAK_TFLOPS,
            "unit": "TFLOP/s", "frac": achieved / FP64_MFMA_PEAK_TFLOPS, "traffic": traffic,
            "traffic_unit": "bytes per launch (PMC, profiles/trsm_traffic.json)",
            "algorithmic_flops_per_launch": float(M_rows) * N * N / nb, "launches_per_step": nb * passes,
            "avg_launch_ms": avg_launch_ms}


def clock_during_step(D_, ctx, step, ms_per_step):
    """shader clock while one more `step` runs (include/robo_hip_diag.h: sampler waves on a second stream) -> the block
    for `roofline.shader_clock_under_kernel`.  EVERY rank calls it (a step of a sharded run contains a collective); the
    peak at that clock is attached on rank 0 by with_clock().  Information only, and only in single-process runs: on the
    HIP runtime a torch.distributed process brings along the sampler's stream did not run beside the
    library's (r03zo: it reported the idle clock), so process-group runs skip it."""
    if D_.mode == "torch":
        return {"skipped": "process-group run; see the single-process line"}
    err = None
    try:
        ctx.clock_sample_begin(max(200, int(0.75 * ms_per_step * 1e3)))
    except Exception as e:
        err = str(e)[:200]
    step()                                   # always, on every rank: the collectives of all ranks must match
    if err is None:
        try:
            return {"mhz": ctx.clock_sample_end()}
        except Exception as e:
            err = str(e)[:200]
    return {"error": err}


def with_clock(roof, clock):
    # information only, UNCALIBRATED: round 4's reading of this sampler gave fractions above 1 of "the peak at that clock"
    # (the sampler waves do not see the clock the matrix pipe runs at), so no fraction is derived from it any more;
    # `frac` (against the nominal 78.6 TFLOP/s) is the only roofline fraction of this line
    if "mhz" in clock:
        clock = dict(clock, note="uncalibrated sampler reading, information only; no fraction is derived from it")
    roof["shader_clock_under_kernel"] = clock
    return roof


# ----------------------------------------------------------------------------------------------------
# headline and config 2: one fitted GP, candidate shard
# ----------------------------------------------------------------------------------------------------
def candidate_shard(args, D_, sharding, scaling=None):
    """this rank's part of the candidate axis -> (candidates here, global index of the first, candidates in the job).
    weak: --m candidates on EVERY rank (the job grows with the ranks); strong: --m candidates in TOTAL, split into
    contiguous shards (SURVEY 8(d)'s headline: one RandomSampling batch, robo/maximizers/random_sampling.py:42-50, G ways)"""
    if (scaling or args.scaling) == "strong":
        b, e = sharding.shard_range(args.m, D_.rank, D_.world)
        return e - b, b, args.m
    return args.m, D_.rank * args.m, args.m * D_.world


def synthetic_candidates(args, D_, sharding, scaling=None):
    """candidates = RandomState(1).rand(M, D).  strong: every rank's shard is a slice of THAT matrix (the
    global argmax is the single-GPU argmax); weak: rank r draws its own batch from RandomState(1 + r)."""
    m_loc, off, m_tot = candidate_shard(args, D_, sharding, scaling)
    if (scaling or args.scaling) == "strong":
        Xc = np.random.RandomState(1).rand(m_tot, args.d)[off:off + m_loc]
    else:
        Xc = np.random.RandomState(1 + D_.rank).rand(m_loc, args.d)
    return np.ascontiguousarray(Xc), m_loc, off, m_tot


def run_headline(args, D_, _lib, sharding):
    rank, world = D_.rank, D_.world
    ctx = _lib.Context(D_.device_ordinal())
    N, D = args.n, args.d
    X, y, theta, _ = synthetic(N, D, 1, rank)
    Xc, M, offset, M_total = synthetic_candidates(args, D_, sharding)
    mean_c, eta = float(np.mean(y)), float(y.min())
    gp = _lib.DeviceGP(ctx, "matern52", N, D)
    gp.set_data(X, y)
    cand = _lib.Candidates(ctx, Xc)          # candidates resident in HBM before the timed region
    comm = D_.make_comm(_lib, ctx)
    ranks = D_.ranks_block(ctx)

    def evaluate_on(cand_, offset_):
        """one pass over this rank's candidates -> the global (max, argmax): with a communicator posterior, EI, local
        argmax, the all-gather of the per-rank incumbents and the cross-rank tie-break are ONE library call"""
        if comm is not None:
            _, mx, am, _, _ = comm.acq_sharded(gp, args.acq, 0.0, eta, cand_, offset_)
            return mx, am
        _, mx, am, _ = gp.acq(args.acq, 0.0, eta, cand_, want_values=False)
        return mx, am

    def evaluate():
        return evaluate_on(cand, offset)

    # ---- GP fit (replicated on every rank) ------------------------------------------------
    fit_ms, fit_phase, fit_ev_ms = [], [], []
    gp.fit(theta, mean_c)                      # first use: allocations
    for _ in range(7):
        t0 = time.perf_counter()
        gp.fit(theta, mean_c)
        fit_ms.append((time.perf_counter() - t0) * 1e3)
    # SURVEY 8(d)'s definition of GP-fit: incl. H2D of X, y, theta and D2H of the log-likelihood
    fit_h2d = []
    for _ in range(7):
        t0 = time.perf_counter()
        gp.set_data(X, y)
        gp.fit(theta, mean_c)
        fit_h2d.append((time.perf_counter() - t0) * 1e3)
    side = {}
    if not args.lean:
        # phase breakdown: separate fits with the library's internal phase events switched on (they are off by
        # default -- the event packets themselves cost a fit ~30 us)
        ctx.set_phase_events(True)
        for _ in range(3):
            t0 = time.perf_counter()
            gp.fit(theta, mean_c)
            fit_ev_ms.append((time.perf_counter() - t0) * 1e3)
            fit_phase.append((ctx.elapsed_ms(20, 21), ctx.elapsed_ms(21, 22), ctx.elapsed_ms(22, 23), ctx.elapsed_ms(19, 21)))
        ctx.set_phase_events(False)
        gram_ms, chol_ms, ll_ms, k1_ms = fit_phase[int(np.argmin(fit_ev_ms))]
        # the MCMC inner loop evaluates half an ensemble of thetas at once (n_hypers = 3 (D + 2) made even = 54 at
        # D = 16 -> 27 per half-step; robo/fmin/bayesian_optimization.py:85-87): one batched pass
        S_half = max(1, (3 * (D + 2) + (3 * (D + 2)) % 2) // 2)
        thetas = theta[None, :] + 0.1 * np.random.RandomState(7).randn(S_half, theta.size)
        gp.loglik_batch(thetas, mean_c)
        batch_runs = []
        for _ in range(5):
            t0 = time.perf_counter()
            gp.loglik_batch(thetas, mean_c)
            batch_runs.append((time.perf_counter() - t0) * 1e3)
        batch_ms = float(np.median(batch_runs))
        gp.grad_loglik(theta, mean_c)
        t0 = time.perf_counter()
        gp.grad_loglik(theta, mean_c)
        grad_ms = (time.perf_counter() - t0) * 1e3
        grad_dev_ms = ctx.elapsed_ms(28, 29)
        gp.fit(theta, mean_c)          # the batch call leaves the GP unfitted
        k1_bytes = 8.0 * N * (N + 1) / 2 + 8.0 * N * D
        side = {
            "gp_fit_phases_ms": {"gram": gram_ms, "cholesky": chol_ms, "loglik": ll_ms},
            # K1 against the HBM roofline: the gram kernel alone (event slots 19 -> 21); "gram" above also holds the
            # staging of theta and the input scaling
            "k_assembly": {"bytes": k1_bytes, "ms": k1_ms, "GB_per_s": k1_bytes / (k1_ms * 1e-3) / 1e9,
                           "frac_of_8TBps": k1_bytes / (k1_ms * 1e-3) / 8.0e12},
            "cholesky": {"flops": N ** 3 / 3.0, "ms": chol_ms, "TFLOP_per_s": N ** 3 / 3.0 / (chol_ms * 1e-3) / 1e12,
                         "frac_of_mfma_peak": N ** 3 / 3.0 / (chol_ms * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS},
            "gp_fit_batched": {"thetas": S_half, "ms_total": batch_ms, "ms_per_theta": batch_ms / S_half,
                               "ms_per_theta_min": float(np.min(batch_runs)) / S_half, "runs": len(batch_runs),
                               "frac_of_mfma_peak": S_half * N ** 3 / 3.0 / (batch_ms * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS},
            "gp_grad_loglik_ms": {"total_incl_fit": grad_ms, "after_factorisation": grad_dev_ms},
        }

    def step():
        return evaluate()

    # batches of <= 16 384 candidates record the solve's event pair (slots 25 -> 26, the roofline's duration) only on
    # request: two event packets, ~2 us of a >= 0.2 ms step
    ctx.set_phase_events(M <= 16384)
    elapsed, best, trsm_ms = timed_steps(D_, ctx, step, args.steps, args.warmup)
    ctx.set_phase_events(False)
    solve_kernel = cand.solve_kernel()
    clock = clock_during_step(D_, ctx, step, elapsed / args.steps * 1e3)  # every rank: the step holds a collective

    # the OTHER scaling of the same job, in the same run (multi-rank runs only; every rank takes part): the driver's
    # `--gpus N` line then carries both the weak figure (--m candidates per GPU) and SURVEY 8(d)'s strong headline
    # (--m candidates in total, split N ways)
    other = None
    if world > 1:
        o_scaling = "weak" if args.scaling == "strong" else "strong"
        Xo, Mo, off_o, Mo_total = synthetic_candidates(args, D_, sharding, o_scaling)
        cand_o = _lib.Candidates(ctx, Xo)
        el_o, best_o, _ = timed_steps(D_, ctx, lambda: evaluate_on(cand_o, off_o), args.steps, max(1, args.warmup))
        other = {"scaling": o_scaling, "value": Mo_total * args.steps / el_o, "unit": "EI evals/s",
                 "ms_per_step": el_o / args.steps * 1e3, "candidates_total": Mo_total,
                 "candidates_rank0": Mo, "solve_kernel_rank0": cand_o.solve_kernel(), "argmax": list(best_o)}
        cand_o.close()

    # SURVEY 8(d)'s full definition of an "EI eval" (PCIe-inclusive: H2D of the candidate batch into an existing
    # handle, D2H of the result) -- reported next to `value`, which is the resident-input rate
    def step_pcie():
        cand.set_points(Xc)
        return evaluate()

    pcie_steps = max(2, args.steps // 2)
    elapsed_pcie = elapsed_pcie_pinned = None
    if not args.lean:
        elapsed_pcie, _, _ = timed_steps(D_, ctx, step_pcie, pcie_steps, 1)
        # the same with the caller's candidate array page-locked (hipHostRegister on the NumPy buffer: what a caller who
        # reuses its batch buffer would do; pageable memory goes through the runtime's own staging copy)
        pinned = _pin_host(Xc)
        if pinned:
            try:
                elapsed_pcie_pinned, _, _ = timed_steps(D_, ctx, step_pcie, pcie_steps, 1)
            finally:
                _unpin_host(Xc)
    # small candidate batches on the same fitted GP (the reference's default RandomSampling draws 500 candidates): latency,
    # not throughput -- the 16/32-candidate block-row step
    small_ms = {}
    if rank == 0 and not args.lean:
        # the explicit-inverse path (winv.hip): W = L^-1 is built lazily by the first small batch after a fit
        gp.fit(theta, mean_c)
        first_ever = None
        for m_small in (500, 8192):
            cs = _lib.Candidates(ctx, np.random.RandomState(11).rand(m_small, D))
            if m_small == 500:
                # the very first small batch of this GP handle also ALLOCATES the n_pad^2 inverse (hipMalloc of 145 MB) and
                # the handle's small-batch workspace: reported under its own name, then a fresh factor for the figure that
                # holds only the W = L^-1 build (what every later refit pays)
                t0 = time.perf_counter()
                gp.acq(args.acq, 0.0, eta, cs, want_values=False)
                first_ever = (time.perf_counter() - t0) * 1e3
                gp.fit(theta, mean_c)
            t0 = time.perf_counter()
            gp.acq(args.acq, 0.0, eta, cs, want_values=False)
            first = (time.perf_counter() - t0) * 1e3
            ts = []
            for _ in range(5):
                t0 = time.perf_counter()
                gp.acq(args.acq, 0.0, eta, cs, want_values=False)
                ts.append((time.perf_counter() - t0) * 1e3)
            small_ms[str(m_small)] = float(np.min(ts))
            small_ms["%d_kernel" % m_small] = cs.solve_kernel()
            if m_small == 500:
                small_ms["500_first_call_after_fit_incl_inverse_build"] = first
                small_ms["500_first_ever_call_incl_allocations"] = first_ever
            ctx.set_tuning("winv_max", 0)          # the block-row substitution on the same batch, for comparison
            gp.acq(args.acq, 0.0, eta, cs, want_values=False)
            ts = []
            for _ in range(3):
                t0 = time.perf_counter()
                gp.acq(args.acq, 0.0, eta, cs, want_values=False)
                ts.append((time.perf_counter() - t0) * 1e3)
            small_ms["%d_block_row_substitution" % m_small] = float(np.min(ts))
            ctx.set_tuning("winv_max", None)
            cs.close()

    out = None
    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = M_total * args.steps / elapsed
        mb = g_tf = g_mhz = None
        if not args.lean:
            try:
                mb = ctx.microbench_mfma_f64_detail(4000)
            except Exception:
                mb = None
            try:
                g_tf, g_mhz = ctx.microbench_gemm_f64(0, 512, 2048, 3)
            except Exception:
                g_tf = g_mhz = None
        traffic = traffic_src = None
        try:   # PMC-measured HBM bytes per launch for this exact workload (tools/gpu_pmc.sh writes it, with the commit)
            tj = json.load(open(os.path.join(ROOT, "profiles", "trsm_traffic.json")))
            if tj["workload"] == {"n_train": N, "dim": D, "candidates_per_gpu": M} and tj["kernel"] in solve_kernel:
                traffic, traffic_src = tj["bytes_per_launch"], tj.get("commit")
        except Exception:
            pass
        if traffic is None and (N, D) == (1024, 8):
            traffic = config_traffic("c2", "trsm_step_gen_kernel")
        roof = roofline_trsm(ctx, N, M, trsm_ms, kernel=solve_kernel, traffic=traffic)
        roof["traffic_measured_at_commit"] = traffic_src
        if mb is not None:
            roof["mfma_f64_microbench"] = mb
        if g_tf:
            roof["gemm_f64_microbench"] = {"lds_core_tflops": g_tf,
                                           "what": "the LDS-staged 128 x 128 fp64 MFMA core alone (no triangular structure, "
                                                   "no K* generation): the ceiling of this GEMM design on this part"}
        with_clock(roof, clock)
        name = "BASELINE headline" if (N, D) == (4096, 16) else "BASELINE config 2" if (N, D) == (1024, 8) else "custom"
        shard_txt = ("%d uniform candidates per GPU" % M) if args.scaling == "weak" else \
            ("%d uniform candidates in total, contiguous shards of %d..%d per GPU" % (
                M_total, M_total // world, -(-M_total // world)))
        out = {
            "metric": METRIC, "value": value, "unit": "EI evals/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            # which rate `value` is: the bench contract's (inputs resident in HBM when the timed region starts).  SURVEY
            # 8(d)'s host-buffer definition of an EI eval (H2D of the batch + D2H of the result inside) is `pcie_inclusive`
            "value_definition": "resident-input rate: candidates already in HBM; posterior + acquisition + argmax + D2H of "
                                "(max, argmax) per step.  The PCIe-inclusive rate is pcie_inclusive.value, never this field",
            "config": {"workload": "GP Matern-5/2 ARD N=%d D=%d, %s, %s xi=0, fp64, "
                                   "candidate shard per GPU (%s)" % (N, D, shard_txt, args.acq.upper(), name),
                       "n_train": N, "dim": D, "candidates_per_gpu": M, "candidates_total": M_total,
                       "acquisition": args.acq,
                       "parallelism": "candidate-shard x%d, replicated fit" % world},
            "algorithmic_tflops_whole_step": value * flops_ei(N, D) / 1e12,
            # GP-fit = robo_gp_fit wall time for one theta INCLUDING the H2D of X, y, theta and the D2H of
            # the log-likelihood; the data-resident refit (what an MCMC / L-BFGS loop pays per theta) next to it
            "gp_fit_ms": float(np.median(fit_h2d)), "gp_fit_ms_min": float(np.min(fit_h2d)),
            "gp_fit_data_resident_ms": float(np.median(fit_ms)), "gp_fit_data_resident_ms_min": float(np.min(fit_ms)),
            "gp_fit_runs": {"incl_h2d": len(fit_h2d), "data_resident": len(fit_ms)},
            "gp_fit_frac_of_mfma_peak": (N ** 3 / 3.0 + N * (N + 1) / 2.0 * (3 * D + 16) + 2.0 * N * N)
            / (float(np.median(fit_ms)) * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS,
            "argmax": list(best), "roofline": roof, "device": ctx.name,
        }
        out.update(ranks)
        out.update(side)
        if elapsed_pcie is not None:
            # `value` is the resident-input rate (the bench contract: inputs in HBM when the timed region starts); SURVEY
            # 8(d)'s host-buffer definition of an "EI eval" is this block
            out["pcie_inclusive"] = {"value": M_total * pcie_steps / elapsed_pcie, "unit": "EI evals/s",
                                     "ms_per_step": elapsed_pcie / pcie_steps * 1e3,
                                     "what": "SURVEY 8(d) definition: H2D of the %d x %d candidate batch (pageable host "
                                             "memory, into an existing handle) + evaluation + D2H of (max, argmax)" % (M, D)}
            if elapsed_pcie_pinned is not None:
                out["pcie_inclusive"]["pinned_host_memory"] = {
                    "value": M_total * pcie_steps / elapsed_pcie_pinned, "ms_per_step": elapsed_pcie_pinned / pcie_steps * 1e3,
                    "what": "the same with the caller's batch page-locked (hipHostRegister)"}
        if small_ms:
            out["small_batch_latency_ms"] = small_ms
        if other is not None:
            out["other_scaling"] = other
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(N, D, theta, X, y)
        if world == 1 and (N, D) == (4096, 16) and not args.lean:
            try:
                out["bo_iteration"] = bo_iteration()
            except Exception as e:            # noqa: BLE001 -- an extra block; never costs the headline line
                out["bo_iteration"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}
            try:
                out["hyper_inference"] = hyper_inference(not args.no_cpu_baseline)
                small = hyper_inference(False, N=100)
                out["hyper_inference"]["n_train_100"] = {k_: small[k_] for k_ in ("later_iteration_ms", "likelihoods_per_s")}
                # where the configurations live (BASELINE config 3: N = 2048; the headline: N = 4096), the reference's
                # defaults (54 walkers, 200 chain steps per later iteration = 10 854 likelihoods)
                for n_mid in (500, 1000):          # between the BO-typical sizes and the configurations'
                    mid = hyper_inference(False, N=n_mid, first_iteration=False)
                    out["hyper_inference"]["n_train_%d" % n_mid] = {k_: mid[k_] for k_ in (
                        "later_iteration_ms", "likelihoods_per_s", "frac_of_fp64_mfma_peak")}
                for n_big, budget in ((2048, 4.0), (4096, 6.0)):
                    out["hyper_inference"]["n_train_%d" % n_big] = hyper_inference(
                        not args.no_cpu_baseline, N=n_big, first_iteration=(n_big == 2048), cpu_budget_s=budget)
                out["bo_iteration"]["gp_mcmc_n2048"] = bo_iteration_gp_mcmc()
            except Exception as e:            # noqa: BLE001 -- an extra block; never costs the headline line
                out["hyper_inference"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}
    return out


# ----------------------------------------------------------------------------------------------------
# config 3: GP-MCMC marginalisation, hyper-parameter samples sharded over the ranks
# ----------------------------------------------------------------------------------------------------
def run_c3(args, D_, _lib, sharding):
    rank, world = D_.rank, D_.world
    ctx = _lib.Context(D_.device_ordinal())
    N, D, M, S = args.n, args.d, args.m, 50
    X, y, theta, Xc = synthetic(N, D, M, 0)       # every rank sees ALL candidates; the samples are sharded
    thetas = theta[None, :] + 0.3 * np.random.RandomState(2).randn(S, theta.size)
    mean_c, eta = float(np.mean(y)), float(y.min())
    b, e = sharding.shard_range(S, rank, world)
    gps = [_lib.DeviceGP(ctx, "matern52", N, D) for _ in range(e - b)]
    gps[0].set_data(X, y)
    cand = _lib.Candidates(ctx, Xc)
    etas = np.full(e - b, eta)
    fit_s = []
    comm = D_.make_comm(_lib, ctx)
    ranks = D_.ranks_block(ctx)

    def step():
        t0 = time.perf_counter()
        _, st = _lib.fit_batch(gps, thetas[b:e], mean_c)          # S_r factorisations in one batched pass
        assert np.all(st == _lib.OK)
        fit_s.append(time.perf_counter() - t0)
        if comm is not None:
            # partial sums stay on the device: RCCL all-gather (512 KB per rank), rank-ordered sum, argmax in the library
            _, mx, am, _ = comm.acq_marginal_sharded(gps, S, "log_ei", 0.0, etas, cand, want_values=False)
        else:
            _, mx, am, _ = _lib.acq_marginal(gps, "log_ei", 0.0, etas, cand, want_values=False)
        return float(mx), int(am)

    elapsed, best, trsm_ms = timed_steps(D_, ctx, step, args.steps, args.warmup)
    clock = clock_during_step(D_, ctx, step, elapsed / args.steps * 1e3)  # every rank: the step holds a collective
    if rank != 0:
        return None
    ms = elapsed / args.steps * 1e3
    # elapsed_ms(25, 26) brackets the LAST sample's solve of a step
    roof = roofline_trsm(ctx, N, M, trsm_ms, kernel=cand.solve_kernel(), traffic=config_traffic("c3", cand.solve_kernel()))
    with_clock(roof, clock)
    return dict(ranks, **{"metric": METRIC, "value": S * M * args.steps / elapsed, "unit": "LogEI sample-evals/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "BASELINE config 3: 50 hyper-parameter samples (theta + 0.3 randn), N=2048 D=16, "
                                   "marginal LogEI over %d candidates; a step = batched fit of the rank's samples + their "
                                   "posteriors + ordered all-gather sum + argmax" % M,
                       "n_train": N, "dim": D, "candidates": M, "samples": S, "acquisition": "log_ei",
                       "parallelism": "sample-shard x%d (%s)" % (world, "/".join(
                           str(sharding.shard_range(S, r, world)[1] - sharding.shard_range(S, r, world)[0])
                           for r in range(world)))},
            "end_to_end_ms": ms, "fit_batch_ms_rank0": float(np.median(fit_s)) * 1e3,
            "fit_ms_per_sample": float(np.median(fit_s)) * 1e3 / (e - b),
            "algorithmic_tflops_whole_step": S * M * flops_ei(N, D) / (ms * 1e-3) / 1e12,
            "argmax": list(best), "roofline": roof, "device": ctx.name},
                **({} if (world > 1 or args.no_cpu_baseline) else {"cpu_baseline": cpu_baseline_c3(N, D, M, thetas, X, y)}))


# ----------------------------------------------------------------------------------------------------
# config 4: Fabolas kernel, information gain per unit cost, candidate shard
# ----------------------------------------------------------------------------------------------------
def c4_problem(N, D):
    """BASELINE config 4's training data: (X with basis (1 - s)^2 on the fidelity column, y, X with the linear basis for
    the cost model, log cost, theta, mean of y)"""
    rs = np.random.RandomState(3)
    X = np.random.RandomState(0).rand(N, D)
    s = rs.rand(N)
    X[:, -1] = (1.0 - s) ** 2                                     # basis (1 - s)^2 on the fidelity column
    y = np.sinc(X[:, :-1] * 10 - 5).sum(axis=1)
    y = (y - y.mean()) / y.std() + 0.5 * X[:, -1]
    cost = np.log(0.2 + 3.0 * s)
    theta = default_theta(D - 1, extra=2)                          # [amp, D-1 metrics, log_a, log_b, noise]
    Xcost = X.copy()
    Xcost[:, -1] = s                                               # linear basis for the cost model
    return X, y, Xcost, cost, theta, float(np.mean(y))


def c4_ep_state(_lib, gp, D, Nb=50, Np=400):
    """representer points on the s = 1 subspace and the EP state of config 4 -> (zb, lmb, EPState, arrays for the CPU port)"""
    from robo_amd.util import epmgp
    from scipy.stats import norm
    zb = np.random.RandomState(4).rand(Nb, D)
    zb[:, -1] = 0.0
    lmb = np.random.RandomState(5).randn(Nb)
    mu_b, cov_b = gp.predict_cov(zb)
    logP, dMu, dSig, dMM = epmgp.joint_min(mu_b, np.clip(cov_b, np.finfo(float).eps, np.inf), with_derivatives=True)
    W = norm.ppf(np.linspace(1. / (Np + 1), 1 - 1. / (Np + 1), Np))[np.newaxis, :]
    return zb, lmb, _lib.EPState(logP, lmb, W, dMu, dSig, dMM), (logP, lmb, dMu, dSig, dMM), W


def run_c4(args, D_, _lib, sharding):
    rank, world = D_.rank, D_.world
    ctx = _lib.Context(D_.device_ordinal())
    N, D, Nb, Np = args.n, args.d, 50, 400
    X, y, Xcost, cost, theta, mean_c = c4_problem(N, D)
    gp, gc = _lib.DeviceGP(ctx, "fabolas", N, D), _lib.DeviceGP(ctx, "fabolas", N, D)
    gp.set_data(X, y)
    gp.fit(theta, mean_c)
    gc.set_data(Xcost, cost)
    gc.fit(theta, float(np.mean(cost)))
    Xc, M, offset, M_total = synthetic_candidates(args, D_, sharding)
    Xc_cost = Xc.copy()
    Xc[:, -1] = (1.0 - Xc[:, -1]) ** 2
    zb, lmb, ep, ep_arrays, W = c4_ep_state(_lib, gp, D, Nb, Np)
    cand, cand_cost, rep = _lib.Candidates(ctx, Xc), _lib.Candidates(ctx, Xc_cost), _lib.Candidates(ctx, zb)
    sn2 = float(np.exp(theta[-1]))
    comm = D_.make_comm(_lib, ctx)
    ranks = D_.ranks_block(ctx)
    ctx.set_phase_events(True)          # batches <= 16384 record the solve's event pair only on request

    def step():
        # posterior + cross-covariances + entropy change, the cost model's posterior, dH / (exp(log cost) + overhead) and
        # the argmax in ONE library call (robo_ig_eval_per_cost_cand); sharded: + the 32-byte all-gather of the per-rank
        # incumbents and the cross-rank tie-break on the device
        if comm is None:
            _, mx, am = _lib.ig_eval_per_cost(gp, cand, rep, ep, sn2, gc, cand_cost, 0.0, want_values=False)
        elif isinstance(comm, TorchExchange):
            _, mx, am = _lib.ig_eval_per_cost(gp, cand, rep, ep, sn2, gc, cand_cost, 0.0, want_values=False)
            rows = comm.allgather([float(mx), float(am + offset)])
            mx, am = sharding.reduce_argmax((float(r[0]), int(r[1])) for r in rows)
        else:
            _, mx, am, _ = comm.ig_per_cost_sharded(gp, cand, rep, ep, sn2, gc, cand_cost, 0.0, offset)
        return float(mx), int(am)

    elapsed, best, trsm_ms = timed_steps(D_, ctx, step, args.steps, args.warmup)
    clock = clock_during_step(D_, ctx, step, elapsed / args.steps * 1e3)  # every rank: the step holds a collective
    if rank != 0:
        return None
    ms = elapsed / args.steps * 1e3
    out = dict(ranks, **{"metric": METRIC, "value": M_total * args.steps / elapsed, "unit": "information gains/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "BASELINE config 4: Fabolas product kernel N=4096 D=10+1, information gain per unit "
                                   "cost (Nb=50, Np=400, objective + cost GP), %d candidates per GPU" % M,
                       "n_train": N, "dim": D, "candidates_per_gpu": M, "candidates_total": M_total,
                       "acquisition": "information_gain_per_unit_cost",
                       "parallelism": "candidate-shard x%d, replicated fits" % world},
            "argmax": list(best),
            "roofline": with_clock(roofline_trsm(ctx, N, M, trsm_ms, kernel=cand_cost.solve_kernel(),
                                                 traffic=config_traffic("c4", cand_cost.solve_kernel())), clock),
            "device": ctx.name,
            "note": "roofline: the block-row solve of the LAST posterior of a step (the cost model's); at this batch "
                    "size the step is latency-bound, not MFMA-bound (see small_batch_latency_ms of the headline line)"})
    if world == 1 and not args.no_cpu_baseline:
        cb = cpu_baseline_c4(N, D, theta, X, y, Xcost, cost, Xc, Xc_cost, zb, ep_arrays, W, sn2)
        # the sampled candidates double as a live check of the device's values against the restated reference loop
        cpu_vals = np.array(cb.pop("_values"))
        dev_vals, _, _ = _lib.ig_eval_per_cost(gp, cand, rep, ep, sn2, gc, cand_cost, 0.0, want_values=True)
        # (the parity tests' measure for information gains: absolute difference in units of the largest gain)
        cb["max_abs_diff_vs_device_on_the_sample_over_max_gain"] = float(
            np.max(np.abs(dev_vals[:cpu_vals.size] - cpu_vals)) / np.max(np.abs(cpu_vals)))
        out["cpu_baseline"] = cb
    return out


# ----------------------------------------------------------------------------------------------------
# config 5: D = 64, N = 8192, LCB, Sobol candidates, fp32 K-build
# ----------------------------------------------------------------------------------------------------
def run_c5(args, D_, _lib, sharding):
    from scipy.stats import qmc
    rank, world = D_.rank, D_.world
    ctx = _lib.Context(D_.device_ordinal())
    N, D = args.n, args.d
    M, offset, M_total = candidate_shard(args, D_, sharding)
    # one workspace pass for the whole shard (131 072 x 8320 doubles = 8.7 GB of the 288 GB): the HIP-event slots
    # bracket the solve of ONE pass, and the roofline below prices all M rows against it
    n_pad = (N + 1 + 127) // 128 * 128
    if "ROBO_WS_BYTES" not in os.environ:
        ctx.set_tuning("ws_bytes", max(12 << 30, min(M * n_pad * 8 + (1 << 20), 160 << 30)))
    X, y, theta, _ = synthetic(N, D, 1, 0)
    gp = _lib.DeviceGP(ctx, "matern52", N, D)
    gp.set_precision(True)
    gp.set_data(X, y)
    t0 = time.perf_counter()
    gp.fit(theta, float(np.mean(y)))
    fit_ms = (time.perf_counter() - t0) * 1e3
    t0 = time.perf_counter()
    gp.fit(theta, float(np.mean(y)))
    fit_ms = min(fit_ms, (time.perf_counter() - t0) * 1e3)
    # this rank's slice of the 2^20-point scrambled Sobol sequence, generated in HBM from SciPy's direction numbers
    # (bit-identical to qmc.Sobol(d, scramble=True, seed=0).random_base2(20)[rank * M : (rank + 1) * M])
    cand = _lib.Candidates(ctx, m=M, sobol=qmc.Sobol(d=D, scramble=True, seed=0), first=offset)

    comm = D_.make_comm(_lib, ctx)
    ranks = D_.ranks_block(ctx)

    def step():
        if comm is not None:
            _, mx, am, _, _ = comm.acq_sharded(gp, "lcb", 1.0, 0.0, cand, offset)
            return mx, am
        _, mx, am, _ = gp.acq("lcb", 1.0, 0.0, cand, want_values=False)
        return mx, am

    elapsed, best, trsm_ms = timed_steps(D_, ctx, step, args.steps, args.warmup)
    clock = clock_during_step(D_, ctx, step, elapsed / args.steps * 1e3)  # every rank: the step holds a collective
    if rank != 0:
        return None
    ms = elapsed / args.steps * 1e3
    out = dict(ranks, **{"metric": METRIC, "value": M_total * args.steps / elapsed, "unit": "LCB evals/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "f64 (covariance entries f32)", "data": "synthetic",
            "config": {"workload": "BASELINE config 5: N=8192 D=64, LCB kappa=1, %d scrambled-Sobol candidates per GPU "
                                   "(slice of 2^20), fp32 K-build + fp64 Cholesky/solve" % M,
                       "n_train": N, "dim": D, "candidates_per_gpu": M, "candidates_total": M_total, "acquisition": "lcb",
                       "parallelism": "candidate-shard x%d, replicated fit" % world},
            "gp_fit_ms": fit_ms, "algorithmic_tflops_whole_step": M_total * flops_ei(N, D) / (ms * 1e-3) / 1e12,
            "argmax": list(best),
            "roofline": with_clock(roofline_trsm(ctx, N, M, trsm_ms, passes=-(-M // cand.chunk()),
                                                 kernel=cand.solve_kernel(),
                                                 traffic=config_traffic("c5", cand.solve_kernel())), clock),
            "device": ctx.name})
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline_c5(N, D, theta, X, y)
    return out


# ----------------------------------------------------------------------------------------------------
# --launcher inproc: ONE process drives all the GPUs (robo_amd/csrc/multi.hip) -- the form a robo.fmin caller uses
# (robo_amd.fmin.*(n_gpus=G)); same workloads, same line shape as the one-process-per-GPU forms above
# ----------------------------------------------------------------------------------------------------
def run_inproc(args, _lib):
    from scipy.stats import qmc
    devices = [int(d) for d in args.devices.split(",")] if args.devices else list(range(args.gpus))
    if len(devices) != args.gpus:
        raise SystemExit("bench.py: --devices names %d devices, --gpus is %d" % (len(devices), args.gpus))
    ctxs = _lib.contexts_for(devices)
    multi = _lib.multi_for(devices) if len(devices) > 1 else _lib.Multi(ctxs)
    G, ctx0 = multi.n, ctxs[0]
    N, D = args.n, args.d
    info = multi.info()
    ranks = {"ranks": list(range(G)), "devices": info[1], "comm_world": G, "launcher": "inproc",
             "worker_threads": info[2]}
    exchange = ("none between processes: %d contexts of one process, one worker thread per device inside librobo_hip; "
                "candidate shards: host reduce of (max, index, flags) per device; sample shards: peer copies to device %d "
                "+ the rank-ordered sum kernel" % (G, info[1][0]))

    def timed(step, steps, warmup, ev_ctx=ctx0):
        for _ in range(warmup):
            step()
        for c in ctxs:
            c.synchronize()
        t0 = time.perf_counter()
        last, trsm = None, 0.0
        for _ in range(steps):
            last = step()
            trsm += ev_ctx.elapsed_ms(25, 26)
        for c in ctxs:
            c.synchronize()
        return time.perf_counter() - t0, last, trsm / steps

    def shard_sizes(total):
        return [_lib.shard_range(total, g, G)[1] - _lib.shard_range(total, g, G)[0] for g in range(G)]

    base = {"metric": METRIC, "n_gpus": G, "steps": args.steps, "warmup": args.warmup, "higher_is_better": True,
            "vs_baseline": None, "data": "synthetic", "device": ctx0.name, "exchange": exchange}
    base.update(ranks)

    if args.config in ("headline", "c2", "c5"):
        kind, par, fp32 = ("lcb", 1.0, True) if args.config == "c5" else (args.acq, 0.0, False)
        X, y, theta, _ = synthetic(N, D, 1, 0)
        mean_c, eta = float(np.mean(y)), float(y.min())
        gps = [_lib.DeviceGP(c, "matern52", N, D) for c in ctxs]
        for g in gps:
            g.set_precision(fp32)
        fit_ms = []
        for _ in range(4):
            t0 = time.perf_counter()
            multi.set_data(gps, X, y)
            multi.fit(gps, theta, mean_c)          # replicated on every device, all devices at once
            fit_ms.append((time.perf_counter() - t0) * 1e3)

        def shards_for(scaling):
            total = args.m if scaling == "strong" else args.m * G
            parts, offs = [], []
            for g, c in enumerate(ctxs):
                b, e = _lib.shard_range(total, g, G) if scaling == "strong" else (g * args.m, (g + 1) * args.m)
                offs.append(b)
                if args.config == "c5":
                    n_pad = (N + 1 + 127) // 128 * 128
                    if "ROBO_WS_BYTES" not in os.environ:
                        c.set_tuning("ws_bytes", max(12 << 30, min((e - b) * n_pad * 8 + (1 << 20), 160 << 30)))
                    parts.append(_lib.Candidates(c, m=e - b, sobol=qmc.Sobol(d=D, scramble=True, seed=0), first=b))
                elif scaling == "strong":
                    parts.append(_lib.Candidates(c, np.ascontiguousarray(np.random.RandomState(1).rand(total, D)[b:e])))
                else:
                    parts.append(_lib.Candidates(c, np.random.RandomState(1 + g).rand(args.m, D)))
            return _lib.CandidateShards(parts, offs), total

        results = {}
        for scaling in ([args.scaling] + (["weak" if args.scaling == "strong" else "strong"] if G > 1 else [])):
            shards, total = shards_for(scaling)
            for c in ctxs:
                c.set_phase_events(total // G <= 16384)

            def step(shards=shards):
                _, mx, am, _, _ = multi.acq(gps, kind, par, eta, shards)
                return mx, am
            el, best, trsm_ms = timed(step, args.steps, args.warmup)
            results[scaling] = {"scaling": scaling, "value": total * args.steps / el, "ms_per_step": el / args.steps * 1e3,
                                "candidates_total": total, "candidates_per_device": shard_sizes(total), "argmax": list(best),
                                "trsm_ms": trsm_ms, "solve_kernel": shards.shards[0].solve_kernel(),
                                "chunk": shards.shards[0].chunk()}
            shards.close()
        for c in ctxs:
            c.set_phase_events(False)
        main_r = results[args.scaling]
        M0 = main_r["candidates_per_device"][0]
        roof = roofline_trsm(ctx0, N, M0, main_r["trsm_ms"], passes=max(1, -(-M0 // max(main_r["chunk"], 1))),
                             kernel=main_r["solve_kernel"])
        unit = "LCB evals/s" if args.config == "c5" else "EI evals/s"
        out = dict(base, value=main_r["value"], unit=unit, ms_per_step=main_r["ms_per_step"], scaling=args.scaling,
                   dtype="f64 (covariance entries f32)" if fp32 else "f64",
                   config={"workload": "GP Matern-5/2 ARD N=%d D=%d, %d candidates in total over %d devices of ONE process, %s, "
                                       "%s" % (N, D, main_r["candidates_total"], G, kind.upper(),
                                               "fp32 K-build + fp64 Cholesky/solve, scrambled-Sobol candidates"
                                               if fp32 else "fp64"),
                           "n_train": N, "dim": D, "candidates_per_gpu": M0, "candidates_total": main_r["candidates_total"],
                           "acquisition": kind, "parallelism": "candidate-shard x%d, replicated fit, single process" % G},
                   algorithmic_tflops_whole_step=main_r["value"] * flops_ei(N, D) / 1e12,
                   gp_fit_ms=float(np.median(fit_ms)), gp_fit_ms_min=float(np.min(fit_ms)),
                   gp_fit_what="robo_gp_set_data_multi + robo_gp_fit_multi: the same fit on all %d devices at once, incl. H2D" % G,
                   argmax=main_r["argmax"], roofline=roof)
        other = [r for k, r in results.items() if k != args.scaling]
        if other:
            o = other[0]
            out["other_scaling"] = {"scaling": o["scaling"], "value": o["value"], "unit": unit, "ms_per_step": o["ms_per_step"],
                                    "candidates_total": o["candidates_total"], "argmax": o["argmax"]}
        return out

    if args.config == "c3":
        S, M = 50, args.m
        X, y, theta, Xc = synthetic(N, D, M, 0)
        thetas = theta[None, :] + 0.3 * np.random.RandomState(2).randn(S, theta.size)
        mean_c, eta = float(np.mean(y)), float(y.min())
        groups = []
        for g, c in enumerate(ctxs):
            b, e = _lib.shard_range(S, g, G)
            grp = [_lib.DeviceGP(c, "matern52", N, D) for _ in range(b, e)]
            if grp:
                grp[0].set_data(X, y)
            groups.append(grp)
        cands = [_lib.Candidates(c, Xc) if (groups[g] or g == 0) else None for g, c in enumerate(ctxs)]
        etas = [np.full(len(grp), eta) for grp in groups]
        fit_s = []

        def step():
            t0 = time.perf_counter()
            _, st = multi.fit_batch(groups, thetas, mean_c)       # every device batch-fits ITS samples, all at once
            assert np.all(st == _lib.OK)
            fit_s.append(time.perf_counter() - t0)
            _, mx, am, _ = multi.acq_marginal(groups, "log_ei", 0.0, etas, cands, want_values=False)
            return float(mx), int(am)
        el, best, trsm_ms = timed(step, args.steps, args.warmup)
        ms = el / args.steps * 1e3
        return dict(base, value=S * M * args.steps / el, unit="LogEI sample-evals/s", ms_per_step=ms, scaling="strong", dtype="f64",
                    config={"workload": "BASELINE config 3: 50 hyper-parameter samples, N=%d D=%d, marginal LogEI over %d "
                                        "candidates; a step = batched fits of every device's samples + their posteriors + peer "
                                        "copies + ordered sum + argmax, ONE process" % (N, D, M),
                            "n_train": N, "dim": D, "candidates": M, "samples": S, "acquisition": "log_ei",
                            "parallelism": "sample-shard x%d (%s), single process" % (G, "/".join(str(len(g)) for g in groups))},
                    end_to_end_ms=ms, fit_batch_ms=float(np.median(fit_s)) * 1e3,
                    algorithmic_tflops_whole_step=S * M * flops_ei(N, D) / (ms * 1e-3) / 1e12, argmax=list(best),
                    roofline=roofline_trsm(ctx0, N, M, trsm_ms, kernel=cands[0].solve_kernel()))

    # config 4: information gain per unit cost, candidate shard
    X, y, Xcost, cost, theta, mean_c = c4_problem(N, D)
    gps, gcs = [_lib.DeviceGP(c, "fabolas", N, D) for c in ctxs], [_lib.DeviceGP(c, "fabolas", N, D) for c in ctxs]
    multi.set_data(gps, X, y)
    multi.fit(gps, theta, mean_c)
    multi.set_data(gcs, Xcost, cost)
    multi.fit(gcs, theta, float(np.mean(cost)))
    zb, lmb, ep, _, _ = c4_ep_state(_lib, gps[0], D)
    total = args.m if args.scaling == "strong" else args.m * G
    Xc_cost = np.random.RandomState(1).rand(total, D)
    Xc = Xc_cost.copy()
    Xc[:, -1] = (1.0 - Xc[:, -1]) ** 2
    shards, cshards = _lib.CandidateShards.split(ctxs, Xc), _lib.CandidateShards.split(ctxs, Xc_cost)
    reps = [_lib.Candidates(c, zb) for c in ctxs]
    sn2 = float(np.exp(theta[-1]))
    for c in ctxs:
        c.set_phase_events(True)

    def step():
        _, mx, am, _ = multi.ig_per_cost(gps, shards, reps, ep, sn2, gcs, cshards, 0.0)
        return float(mx), int(am)
    el, best, trsm_ms = timed(step, args.steps, args.warmup)
    ms = el / args.steps * 1e3
    return dict(base, value=total * args.steps / el, unit="information gains/s", ms_per_step=ms, scaling=args.scaling, dtype="f64",
                config={"workload": "BASELINE config 4: Fabolas product kernel N=%d D=%d, information gain per unit cost (Nb=50, "
                                    "Np=400, objective + cost GP), %d candidates over %d devices of ONE process" % (N, D, total, G),
                        "n_train": N, "dim": D, "candidates_per_gpu": shard_sizes(total)[0], "candidates_total": total,
                        "acquisition": "information_gain_per_unit_cost",
                        "parallelism": "candidate-shard x%d, replicated fits, single process" % G},
                argmax=list(best),
                roofline=roofline_trsm(ctx0, N, shard_sizes(total)[0], trsm_ms, kernel=cshards.shards[0].solve_kernel()))


CONFIG_DEFAULTS = {"headline": (4096, 16, 65536), "c2": (1024, 8, 65536), "c3": (2048, 16, 65536),
                   "c4": (4096, 11, 8192), "c5": (8192, 64, 131072)}


def config_args(args, cfg, lean=False):
    """the argument set `--config cfg` would run with (its sizes, its sustained step counts), derived from a parsed one"""
    import copy
    a = copy.copy(args)
    a.config = cfg
    a.n, a.d, a.m = CONFIG_DEFAULTS[cfg]
    a.steps = {"c2": 100, "c4": 50}.get(cfg, 5)
    a.warmup = 5 if cfg in ("c2", "c4") else 2
    a.scaling = "strong" if cfg == "c3" else "weak"
    if lean:
        a.lean = True
        a.no_cpu_baseline = True
    return a


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None,
                    help="timed steps (default 5; config c2: 100, c4: 50 -- their steps last 1.3 / 5.3 ms, and a 5-step run "
                         "ends before the part has left its idle power state)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed steps in front (default 2; c2 / c4: 5)")
    ap.add_argument("--config", default="headline", choices=["headline", "c2", "c3", "c4", "c5", "all"],
                    help="all: every configuration in turn, ONE JSON array of their lines")
    ap.add_argument("--n", type=int, default=None)
    ap.add_argument("--d", type=int, default=None)
    ap.add_argument("--m", type=int, default=None, help="candidates per GPU (config 3: candidates in total)")
    ap.add_argument("--acq", default="ei")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--lib", default=None, help="alternative build of librobo_hip.so (A/B runs of kernel variants)")
    ap.add_argument("--scaling", default=None, choices=["weak", "strong"],
                    help="weak (default): --m candidates per GPU; strong: --m candidates in TOTAL, split over the ranks "
                         "(SURVEY 8(d)'s headline; config 3 shards its 50 samples and is always strong)")
    ap.add_argument("--launcher", default=None, choices=["inproc"],
                    help="inproc: ONE process drives all --gpus devices (robo_amd/csrc/multi.hip), the form robo_amd.fmin.*("
                         "n_gpus=G) uses; default: one process per GPU (this script starts them, or torchrun did)")
    ap.add_argument("--devices", default=None, help="inproc: comma-separated HIP device ids (default 0..gpus-1; a device "
                                                    "may repeat: several contexts on it)")
    ap.add_argument("--lean", action="store_true",
                    help="the timed region and the fit only: no side measurements (phases, micro-benchmarks, small batches)")
    args = ap.parse_args()
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is not None and int(env_world) != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but the launcher started WORLD_SIZE=%s ranks; refusing to report a line "
                         "for a job that is not the one asked for\n" % (args.gpus, env_world))
        sys.exit(2)
    if args.launcher == "inproc" and env_world is not None and int(env_world) > 1:
        ap.error("--launcher inproc is ONE process; it cannot run under a multi-rank launcher")
    if env_world is None and args.gpus > 1 and args.launcher != "inproc":
        # no launcher around this process: be the launcher (one rank process per GPU), pass rank 0's line through
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    short_steps = args.config in ("c2", "c4")
    if args.steps is None:
        args.steps = {"c2": 100, "c4": 50}.get(args.config, 5)
    if args.warmup is None:
        args.warmup = 5 if short_steps else 2
    if args.config == "c3":
        args.scaling = "strong"
    args.scaling = args.scaling or "weak"
    defaults = CONFIG_DEFAULTS["headline" if args.config == "all" else args.config]
    if args.scaling == "strong" and args.config in ("c4", "c5"):      # BASELINE's totals: 8 x 8192, 8 x 2^17
        defaults = defaults[:2] + (defaults[2] * 8,)
    args.n = args.n or defaults[0]
    args.d = args.d or defaults[1]
    args.m = args.m or defaults[2]

    if args.launcher == "inproc":
        from robo_amd import _lib
        if args.lib:
            _lib.use_library(os.path.abspath(args.lib))
        sys.stdout.write(json.dumps(run_inproc(args, _lib)) + "\n")
        sys.stdout.flush()
        return
    D_ = Dist()
    from robo_amd import _lib, sharding
    if args.lib:
        _lib.use_library(os.path.abspath(args.lib))
    try:
        # the diagnostics library (micro-benchmarks of the roofline block) is loaded BEFORE the first kernel launch:
        # under rocprofv3 a code object that appears after thousands of dispatches crashed the profiler's launch
        # interception (r02v: SIGSEGV at the first mfma_bench_kernel launch)
        _lib.diag()
    except Exception:
        pass
    runners = {"headline": run_headline, "c2": run_headline, "c3": run_c3, "c4": run_c4, "c5": run_c5}
    if args.config == "all":
        # every configuration of BASELINE.json that fits one GPU, each its own complete line (own cpu_baseline unless
        # --no-cpu-baseline), printed as ONE JSON array: `python bench.py --gpus 1 --config all`
        lines = []
        for cfg in ("headline", "c2", "c3", "c4", "c5"):
            o = runners[cfg](config_args(args, cfg), D_, _lib, sharding)
            if D_.rank == 0:
                o.setdefault("exchange", D_.exchange or "none (single process, no communicator)")
                lines.append(o)
        if D_.rank == 0:
            sys.stdout.write(json.dumps(lines) + "\n")
            sys.stdout.flush()
        D_.close()
        return
    out = runners[args.config](args, D_, _lib, sharding)
    if D_.rank == 0 and D_.world == 1 and args.config == "headline" and (args.n, args.d) == (4096, 16) and not args.lean \
            and os.environ.get("ROBO_BENCH_NO_CONFIGS") != "1":
        # the DEFAULT run (what the round's driver times) also carries BASELINE's configurations 2-5 at their one-GPU
        # sizes, each measured by its own runner in this process (sustained step counts as in `--config cN`; no CPU legs
        # here -- `--config cN` / `--config all` print the full lines with their cpu_baseline blocks)
        out["configs"] = {}
        for cfg in ("c2", "c3", "c4", "c5"):
            try:
                t0 = time.perf_counter()
                o = runners[cfg](config_args(args, cfg, lean=True), D_, _lib, sharding)
                out["configs"][cfg] = {"workload": o["config"]["workload"], "value": o["value"], "unit": o["unit"],
                                       "ms_per_step": o["ms_per_step"], "steps": o["steps"], "warmup": o["warmup"],
                                       "dtype": o["dtype"], "argmax": o.get("argmax"),
                                       "roofline": {k_: o.get("roofline", {}).get(k_) for k_ in (
                                           "kernel", "bound", "achieved", "peak", "unit", "frac", "traffic")},
                                       "wall_s_incl_setup": time.perf_counter() - t0}
            except Exception as e:            # noqa: BLE001 -- an extra block; never costs the headline line
                out["configs"][cfg] = {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}
    if D_.rank == 0:
        out.setdefault("exchange", D_.exchange or "none (single process, no communicator)")
        D_.emit(out)
    D_.close()


if __name__ == "__main__":
    main()

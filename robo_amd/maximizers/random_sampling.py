"""Acquisition maximisation by candidate sampling -- the batched caller of the hot path.

``RandomSampling`` keeps the constructor and the candidate recipe of
robo/maximizers/random_sampling.py:7-52: 70 % uniform points (``init_random_uniform``
called WITHOUT the maximiser's rng, :38-39) and 30 % drawn from N(incumbent, 0.1) with the
GLOBAL ``np.random`` and an absolute sigma (:43-45), clipped to the box; one batched
``acq(X)`` call; ``X[y.argmax()]``.  ``n_samples`` may be far larger than the reference's
500: the whole batch is evaluated by one device call and, when the acquisition function
offers ``argmax``, only the winning index crosses PCIe.
"""
import numpy as np

from robo_amd.initial_design import init_random_uniform


class BaseMaximizer(object):
    """robo/maximizers/base_maximizer.py:5-32."""

    def __init__(self, objective_function, lower, upper, rng=None):
        self.lower = lower
        self.upper = upper
        self.objective_func = objective_function
        self.rng = np.random.RandomState(np.random.randint(10000)) if rng is None else rng

    def maximize(self):
        pass


class RandomSampling(BaseMaximizer):

    def __init__(self, objective_function, lower, upper, n_samples=500, rng=None, device_argmax=True, shard=False):
        self.n_samples = n_samples
        self.device_argmax = device_argmax
        # shard=True (explicit opt-in, like GaussianProcessMCMC.sample_shard): with one process per GPU every rank
        # evaluates its slice of the SAME candidate matrix and the per-shard incumbents are exchanged.  All ranks
        # must then run the same BO loop with the same seeds and call maximize() in lock step; that the candidate
        # matrices agree is checked before the exchange.  An initialised process group alone does NOT switch this on.
        self.shard = shard
        super(RandomSampling, self).__init__(objective_function, lower, upper, rng)

    def candidates(self):
        n_uniform = int(self.n_samples * .7)
        n_local = int(self.n_samples * 0.3)
        rand = init_random_uniform(self.lower, self.upper, n_uniform)
        loc = self.objective_func.model.get_incumbent()[0]
        scale = np.ones([self.lower.shape[0]]) * 0.1
        # the reference draws np.random.normal(loc, scale) once per point in a list comprehension (:44) from the GLOBAL
        # stream; one (n_local, D) call takes the same numbers from that stream in the same order (bit-identical, also
        # across the cached second Gaussian of odd D: tests/test_host_logic.py); the clip is applied once to the block
        local = np.clip(np.random.normal(loc, scale, (n_local, self.lower.shape[0])), self.lower, self.upper)
        return np.concatenate((rand, local), axis=0)

    def maximize(self):
        X = self.candidates()
        if self.shard:
            from robo_amd import sharding
            if sharding.dist_info()[2] > 1:
                # one process per GPU: same seeds, hence the same X, on every rank; each evaluates its slice
                sharding.assert_replicated("RandomSampling candidates", [X.shape[0], X.shape[1], float(X.sum()),
                                                                         float(X[0, 0]), float(X[-1, -1])])
                return X[sharding.sharded_argmax(self.objective_func, X)]
        if self.device_argmax and hasattr(self.objective_func, "argmax"):
            return X[self.objective_func.argmax(X)]
        y = self.objective_func(X)
        return X[y.argmax()]


class DeviceRandomSampling(BaseMaximizer):
    """Large-M variant of :class:`RandomSampling` whose candidates never exist on the host.

    Same recipe (70 % uniform over the box, 30 % N(incumbent, 0.1) clipped,
    robo/maximizers/random_sampling.py:38-47) generated ON THE DEVICE with a counter-based
    generator, evaluated by one fused acquisition call, and only the winning row is copied back
    (SURVEY.md section 8f rank 2).  Needs a robo_amd model (GaussianProcess or a
    MarginalizationGPMCMC over them) because the candidates live in the model's normalised input
    space; the random stream is Philox, so the sequence differs from the reference's NumPy one.
    """

    def __init__(self, objective_function, lower, upper, n_samples=65536, rng=None, shard=False):
        super(DeviceRandomSampling, self).__init__(objective_function, lower, upper, rng)
        self.n_samples = int(n_samples)
        # shard=True: explicit opt-in to the candidate shard (see RandomSampling); rank r then draws rows [b, e) of the
        # recipe from ITS OWN Philox stream (seed + 7919 r), so the sharded candidate set is a different (equally
        # distributed) sample than the single-process one
        self.shard = shard

    def maximize(self):
        from robo_amd import _lib
        acq = self.objective_func
        model = acq.model
        sub = model.models[0] if hasattr(model, "models") and len(model.models) > 0 else model
        if not getattr(sub, "normalize_input", False) or not hasattr(sub, "gp"):
            raise TypeError("DeviceRandomSampling needs a robo_amd GP model with normalize_input=True")
        lower, upper = np.asarray(sub.lower, dtype=np.float64), np.asarray(sub.upper, dtype=np.float64)
        inc = np.asarray(model.get_incumbent()[0], dtype=np.float64)
        loc = (inc - lower) / (upper - lower)
        scale = 0.1 / (upper - lower)
        seed = int(self.rng.randint(0, 2 ** 31 - 1))
        n_uniform = int(self.n_samples * .7)
        if getattr(sub, "devices", None) and sub is model:
            # single-process multi-GPU (``GaussianProcess(devices=...)``): slot g generates and scores rows [b, e) of the
            # recipe on ITS device -- the same per-shard streams as the one-process-per-GPU form below --, the incumbents
            # are reduced inside the library call and only the winning row comes back
            multi = sub._multi()
            parts, offsets = [], []
            for g, ctx in enumerate(multi.ctxs):
                b, e = _lib.shard_range(self.n_samples, g, multi.n)
                offsets.append(b)
                parts.append(_lib.Candidates(ctx, m=e - b, seed=seed + 7919 * g,
                                             n_uniform=min(max(n_uniform - b, 0), e - b), loc=loc, scale=scale) if e > b else None)
            shards = _lib.CandidateShards(parts, offsets)
            try:
                return lower + (upper - lower) * shards.point(acq.argmax(shards))
            finally:
                shards.close()
        from robo_amd import sharding
        _, rank, world = sharding.dist_info() if self.shard else (None, 0, 1)
        if world > 1:
            sharding.assert_replicated("DeviceRandomSampling (n_samples, seed, incumbent)",
                                       [self.n_samples, seed] + [float(v) for v in loc])
        # candidate shard (one process per GPU, same rng seed everywhere): rank r generates and evaluates rows
        # [b, e) of the recipe -- its own Philox stream, the 70 % / 30 % split kept globally -- and only the
        # per-shard incumbent (16 B) and the winning point (D doubles) are exchanged
        b, e = sharding.shard_range(self.n_samples, rank, world)
        cand = _lib.Candidates(sub.gp.ctx, m=max(e - b, 1), seed=seed + 7919 * rank,
                               n_uniform=min(max(n_uniform - b, 0), max(e - b, 1)), loc=loc, scale=scale)
        try:
            best = acq.argmax(cand)
            point = cand.point(best)
            if world > 1:
                _, win = sharding.allgather_argmax(acq.last_max, b + best)
                points = sharding.allgather_rows(point)
                owner = [r for r in range(world) if sharding.shard_range(self.n_samples, r, world)[0] <= win
                         < sharding.shard_range(self.n_samples, r, world)[1]][0]
                point = points[owner]
            return lower + (upper - lower) * point
        finally:
            cand.close()


class DeviceSobolSampling(BaseMaximizer):
    """Acquisition maximisation over the first ``n_samples`` points of a scrambled Sobol' sequence in the box,
    generated on the device (BASELINE config 5: 2^20 candidates in 64 dimensions -- 537 MB that never exist on the
    host).  The sequence is SciPy's ``qmc.Sobol(d, scramble=True, seed=seed)`` bit for bit (its direction numbers
    and digital shift are handed to robo_cand_create_sobol); with one process per GPU every rank generates and
    evaluates its own contiguous slice and only the per-shard incumbent and the winning point are exchanged.
    Not in the reference (which has no Sobol sampler); same maximiser protocol as RandomSampling."""

    def __init__(self, objective_function, lower, upper, n_samples=2 ** 16, seed=0, rng=None, shard=False):
        super(DeviceSobolSampling, self).__init__(objective_function, lower, upper, rng)
        self.n_samples = int(n_samples)
        self.seed = seed
        self.shard = shard      # explicit opt-in to the candidate shard (slices of ONE sequence: same result as unsharded)

    def maximize(self):
        from scipy.stats import qmc
        from robo_amd import _lib, sharding
        acq = self.objective_func
        model = acq.model
        sub = model.models[0] if hasattr(model, "models") and len(model.models) > 0 else model
        if not getattr(sub, "normalize_input", False) or not hasattr(sub, "gp"):
            raise TypeError("DeviceSobolSampling needs a robo_amd GP model with normalize_input=True")
        lower, upper = np.asarray(sub.lower, dtype=np.float64), np.asarray(sub.upper, dtype=np.float64)
        eng = qmc.Sobol(d=lower.shape[0], scramble=True, seed=self.seed)
        if getattr(sub, "devices", None) and sub is model:
            # single-process multi-GPU: slot g generates and scores points [b, e) of the ONE sequence on its device
            multi = sub._multi()
            parts, offsets = [], []
            for g, ctx in enumerate(multi.ctxs):
                b, e = _lib.shard_range(self.n_samples, g, multi.n)
                offsets.append(b)
                parts.append(_lib.Candidates(ctx, m=e - b, sobol=eng, first=b) if e > b else None)
            shards = _lib.CandidateShards(parts, offsets)
            try:
                return lower + (upper - lower) * shards.point(acq.argmax(shards))
            finally:
                shards.close()
        _, rank, world = sharding.dist_info() if self.shard else (None, 0, 1)
        if world > 1:
            sharding.assert_replicated("DeviceSobolSampling (n_samples, seed)", [self.n_samples, int(self.seed)])
        b, e = sharding.shard_range(self.n_samples, rank, world)
        cand = _lib.Candidates(sub.gp.ctx, m=max(e - b, 1), sobol=eng, first=b)
        try:
            best = acq.argmax(cand)
            point = cand.point(best)
            if world > 1:
                _, win = sharding.allgather_argmax(acq.last_max, b + best)
                points = sharding.allgather_rows(point)
                owner = [r for r in range(world) if sharding.shard_range(self.n_samples, r, world)[0] <= win
                         < sharding.shard_range(self.n_samples, r, world)[1]][0]
                point = points[owner]
            return lower + (upper - lower) * point
        finally:
            cand.close()

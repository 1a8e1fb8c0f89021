"""GP with MCMC-marginalised hyper-parameters on the MI355X hot path.

Same constructor, attributes (``models``, ``n_hypers``, ``chain_length``, ``burnin_steps``,
``burned``, ``p0``, ``hypers``) and semantics as robo/models/gaussian_process_mcmc.py:16-269.
The ensemble sampler evaluates each half-ensemble's log-likelihoods in ONE device call
(robo_gp_loglik_batch) instead of one george fit per walker; every hyper-parameter sample
then gets its own device-resident GaussianProcess (``self.models``), which is what
MarginalizationGPMCMC iterates over (marginalization.py:34-46,118).
"""
import logging
import os
from copy import deepcopy

import numpy as np

from robo_amd import _lib
from robo_amd.models.base_model import BaseModel
from robo_amd.models.gaussian_process import GaussianProcess
from robo_amd.util import normalization
from robo_amd.util.ensemble_sampler import EnsembleSampler

logger = logging.getLogger(__name__)


class GaussianProcessMCMC(BaseModel):

    def __init__(self, kernel, prior=None, n_hypers=20, chain_length=2000, burnin_steps=2000,
                 normalize_output=False, normalize_input=True, rng=None, lower=None, upper=None, noise=-8,
                 device=None, devices=None):
        if rng is None:
            self.rng = np.random.RandomState(np.random.randint(0, 10000))
        else:
            self.rng = rng
        self.kernel = kernel
        self.prior = prior
        self.noise = noise              # log scale here (gaussian_process_mcmc.py:20,144-147)
        self.n_hypers = n_hypers
        self.chain_length = chain_length
        self.burned = False
        self.burnin_steps = burnin_steps
        self.models = []
        self.normalize_output = normalize_output
        self.normalize_input = normalize_input
        self.X = None
        self.y = None
        self.is_trained = False
        self.lower = lower
        self.upper = upper
        self.device = device
        # devices = [d0, d1, ...] (not in the reference): single-process multi-GPU.  The hyper-parameter samples are
        # split contiguously over the listed devices (BASELINE config 3: 50 samples over 4 GPUs -> 13/13/12/12): sample s
        # is fitted, kept and evaluated on its device (robo_gp_fit_batch_multi, robo_acq_eval_marginal_cand_multi,
        # robo_gp_predict_mixture_cand_multi); the walkers of an ensemble half-step are split the same way once a
        # likelihood is expensive enough to pay for a host round trip per half-step (walker_shard_min_n points).
        self.devices = _lib.resolve_devices(devices)
        self.walker_shard_min_n = 1024
        self.walker_gps = []            # devices[1:]: data-holding handles for the walkers' likelihoods
        self.gp = None                  # scratch device GP for the likelihood evaluations
        # one process per GPU: fit only this rank's shard of the hyper-parameter samples (the MCMC itself is
        # replicated -- same seeds, same chain on every rank); used with MarginalizationGPMCMC.sample_shard
        self.sample_shard = False

    def _multi(self):
        return _lib.multi_for(self.devices) if self.devices else None

    def _ensure_gp(self, n, dim):
        if self.gp is None or self.gp.dim != dim or self.gp.n_max < n or self.gp.kind != self.kernel.kind:
            for g in [self.gp] + list(self.walker_gps):
                if g is not None:
                    g.close()
            self.walker_gps = []
            cap = max(127, int(n)) if self.gp is None else max(int(n), 2 * self.gp.n_max)
            ctx = self._multi().ctxs[0] if self.devices else _lib.default_context(self.device)
            self.gp = _lib.DeviceGP(ctx, self.kernel.kind, cap, dim, fixed_head=self.kernel.fixed_head())
        return self.gp

    def _walker_shard(self):
        return bool(self.devices) and self.X is not None and self.X.shape[0] >= self.walker_shard_min_n

    def _slot_of(self, i, n_samples):
        """device slot of hyper-parameter sample i (contiguous shards, the first n % G slots hold one more)"""
        for g in range(len(self.devices)):
            b, e = _lib.shard_range(n_samples, g, len(self.devices))
            if b <= i < e:
                return g
        raise IndexError(i)

    def _groups(self):
        """the trained sub-models' device handles grouped by device slot, in sample order"""
        G = len(self.devices)
        groups = [[] for _ in range(G)]
        for i, m in enumerate(self.models):
            groups[self._slot_of(i, len(self.models))].append(m)
        return groups

    def _on_their_slots(self):
        """every trained sub-model's handle lives on the context of ITS device slot (after a pickle round trip the
        sub-models rematerialise on the default context -- GaussianProcess.__getstate__ drops the override -- and the
        multi-device entry points would refuse them: then the plain per-model path is taken instead)"""
        ctxs = self._multi().ctxs
        return all(getattr(m, "gp", None) is not None and m.gp.ctx is ctxs[self._slot_of(i, len(self.models))]
                   for i, m in enumerate(self.models))

    def __deepcopy__(self, memo):
        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            setattr(new, k, None if k == "gp" else ([] if k == "walker_gps" else deepcopy(v, memo)))
        return new

    def __getstate__(self):
        d = dict(self.__dict__)
        d["gp"] = None
        d["walker_gps"] = []
        return d

    # hooks for FabolasGPMCMC -------------------------------------------------------------------
    def _model_inputs(self, X):
        """the inputs the kernel sees (None: standard [0,1] normalisation when enabled)"""
        return None

    def _make_model(self, kernel, noise):
        return GaussianProcess(kernel, normalize_output=self.normalize_output, normalize_input=self.normalize_input,
                               noise=noise, lower=self.lower, upper=self.upper, rng=self.rng, device=self.device)

    @BaseModel._check_shapes_train
    def train(self, X, y, do_optimize=True, **kwargs):
        self.original_X = X
        mapped = self._model_inputs(X)
        if mapped is not None:
            self.X = mapped
        elif self.normalize_input:
            self.X, self.lower, self.upper = normalization.zero_one_normalization(X, self.lower, self.upper)
        else:
            self.X = X
        if self.normalize_output:
            self.y, self.y_mean, self.y_std = normalization.zero_mean_unit_var_normalization(y)
            if self.y_std == 0:
                raise ValueError("Cannot normalize output. All targets have the same value")
        else:
            self.y = y
        self.mean = np.mean(self.y, axis=0)
        gp = self._ensure_gp(self.X.shape[0], self.X.shape[1])
        if self._walker_shard() and (do_optimize or self.walker_gps):
            # (existing walker handles receive EVERY training's data, also a do_optimize=False one: loglikelihood_batch
            # scores through them and must never mix two data sets -- the solver's train_interval > 1, Fabolas)
            if not self.walker_gps:
                self.walker_gps = [_lib.DeviceGP(c, self.kernel.kind, gp.n_max, gp.dim, fixed_head=gp.fixed_head)
                                   for c in self._multi().ctxs[1:]]
            self._multi().set_data([gp] + self.walker_gps, self.X, self.y)
        else:
            gp.set_data(self.X, self.y)

        if do_optimize:
            sampler = EnsembleSampler(self.n_hypers, len(self.kernel) + 1, lnprob_batch=self.loglikelihood_batch,
                                      device_chain=self._device_chain())
            sampler.random_state = self.rng.get_state()
            if not self.burned:
                if self.prior is None:
                    self.p0 = self.rng.rand(self.n_hypers, len(self.kernel) + 1)
                else:
                    self.p0 = self.prior.sample_from_prior(self.n_hypers)
                self.p0, _, _ = sampler.run_mcmc(self.p0, self.burnin_steps, rstate0=self.rng)
                self.burned = True
            pos, _, _ = sampler.run_mcmc(self.p0, self.chain_length, rstate0=self.rng)
            self.p0 = pos
            self.hypers = sampler.chain[:, -1]
        elif getattr(self, "hypers", None) is None or not self._keep_hypers_without_optimize():
            self.hypers = self.kernel[:].tolist()
            self.hypers.append(self.noise)
            self.hypers = [self.hypers]

        # one device-resident GP per hyper-parameter sample; handles of the previous round are
        # reused so a BO run does not re-allocate S factor buffers every iteration
        old = self.models
        self.models = []
        for i, sample in enumerate(self.hypers):
            sample = np.asarray(sample, dtype=np.float64)
            kernel = deepcopy(self.kernel)
            kernel.set_parameter_vector(sample[:-1])
            if i < len(old) and isinstance(old[i], GaussianProcess):
                model = old[i]
                model.kernel = kernel
                model.noise = np.exp(sample[-1])
                model.lower, model.upper = self.lower, self.upper
            else:
                model = self._make_model(kernel, np.exp(sample[-1]))
            if self.devices:
                # sample i lives on the device of its shard; a handle left on another device by an earlier, differently
                # sized set of samples is dropped
                ctx = self._multi().ctxs[self._slot_of(i, len(self.hypers))]
                if model._ctx_override is not ctx:
                    if model.gp is not None:
                        model.gp.close()
                        model.gp = None
                    model._ctx_override = ctx
            self.models.append(model)
        for m in old[len(self.models):]:
            if getattr(m, "gp", None) is not None:
                m.gp.close()
        self._fit_models(X, y)
        self.is_trained = True

    def _fit_models(self, X, y):
        """``model.train(X, y, do_optimize=False)`` for every hyper-parameter sample
        (gaussian_process_mcmc.py:149-164) -- S Cholesky factorisations of the same data.  With more than one
        sample they run as ONE batched device pass that leaves S fitted handles (robo_gp_fit_batch,
        bit-identical to S sequential fits); a sample whose K is not positive definite goes through the
        model's own train(), i.e. the reference's noise x 10 retry (gaussian_process.py:120-122)."""
        models = self.models
        if self.devices and os.environ.get("ROBO_MCMC_SEQUENTIAL_FITS") != "1":
            # single-process multi-GPU: every device batch-fits ITS samples, all devices at once
            groups = self._groups()
            gp_groups = [[m._host_train_raw(X, y) for m in grp] for grp in groups]
            flat = [m for grp in groups for m in grp]
            m0 = flat[0]
            if all(g.n_max >= m0.X.shape[0] for grp in gp_groups for g in grp):
                for grp, gps in zip(groups, gp_groups):
                    if grp:
                        gps[0].set_data(grp[0].X, grp[0].y)
                thetas = np.array([np.append(m.kernel.get_parameter_vector(), np.log(m.noise)) for m in flat])
                _, st = self._multi().fit_batch(gp_groups, thetas, m0.mean)
                for model, theta, status in zip(flat, thetas, st):
                    if status == _lib.OK:
                        model._adopt_fit(theta)
                    else:
                        model.train(X, y, do_optimize=False)
                self._prefetch_inverses(flat)
                return
        if self.sample_shard:
            from robo_amd import sharding
            _, rank, world = sharding.dist_info()
            if world > 1:
                b, e = sharding.shard_range(len(models), rank, world)
                for i, model in enumerate(models):
                    if not b <= i < e:
                        model._host_train_raw(X, y, alloc=False)   # host state only; never evaluated on this rank
                models = models[b:e]
        if len(models) < 2 or os.environ.get("ROBO_MCMC_SEQUENTIAL_FITS") == "1":
            for model in models:
                model.train(X, y, do_optimize=False)
            return
        gps = [model._host_train_raw(X, y) for model in models]
        m0 = models[0]
        if len({id(g) for g in gps}) != len(gps) or any(g.n_max < m0.X.shape[0] for g in gps):
            for model in models:
                model.train(X, y, do_optimize=False)
            return
        gps[0].set_data(m0.X, m0.y)
        thetas = np.array([np.append(m.kernel.get_parameter_vector(), np.log(m.noise)) for m in models])
        _, st = _lib.fit_batch(gps, thetas, m0.mean)
        for model, theta, status in zip(models, thetas, st):
            if status == _lib.OK:
                model._adopt_fit(theta)
            else:
                model.train(X, y, do_optimize=False)
        self._prefetch_inverses(models)

    @staticmethod
    def _prefetch_inverses(models):
        """launch every sample's explicit-inverse build now (asynchronous, best effort): the marginal acquisition over a
        small candidate batch that follows (500 candidates by default) then finds S inverses in place instead of building
        -- and waiting for -- one per sample inside its loop"""
        for m in models:
            try:
                if m.gp is not None:
                    m.gp.prefetch_inverse()
            except Exception as e:        # noqa: BLE001
                logger.warning("prefetch of the explicit inverse factor failed (%s); it is built on first use", e)

    def _keep_hypers_without_optimize(self):
        # FabolasGPMCMC keeps the previous samples when do_optimize=False (fabolas_gp.py:80-84);
        # the plain model resets to the kernel's current vector (gaussian_process_mcmc.py:144-147)
        return hasattr(self, "basis_func")

    # ---- likelihood ------------------------------------------------------------------------------
    def _device_chain(self):
        """The whole ensemble chain on the device (robo_gp_mcmc_run) when the prior is one the library evaluates itself:
        none, exactly DefaultPrior (robo/priors/default_priors.py) or exactly EnvPrior (robo/priors/env_priors.py:8-54,
        FabolasGPMCMC's prior in robo.fmin.fabolas).  Other priors keep the host sampler around the batched likelihood.
        ROBO_MCMC_HOST=1 forces the host sampler (A/B, tests)."""
        if os.environ.get("ROBO_MCMC_HOST") == "1" or self._walker_shard():
            return None                 # (walker shard: the half-ensemble's likelihoods are split over the devices per half-step)
        if self.kernel.fixed_head():
            return None                 # a kernel without an amplitude parameter: the device chain would move the amplitude
        from robo_amd.priors import DefaultPrior, EnvPrior
        if self.prior is None:
            prior = None
        elif type(self.prior) is DefaultPrior:
            pr = self.prior
            prior = (1, [pr.ln_prior.mean, pr.ln_prior.sigma, pr.tophat.min, pr.tophat.max, pr.horseshoe.scale])
        elif type(self.prior) is EnvPrior and 1 + self.prior.n_ls + self.prior.n_lr <= len(self.kernel):
            pr = self.prior
            prior = (2, [pr.ln_prior.mean, pr.ln_prior.sigma, pr.tophat.min, pr.tophat.max, pr.horseshoe.scale,
                         pr.n_ls, pr.n_lr, pr.bayes_lin_prior.mean, pr.bayes_lin_prior.sigma])
        else:
            return None

        def run(p, lnp, n_steps, u_stretch, partner, u_accept, a):
            try:
                return self.gp.mcmc_run(self.mean, prior, p, lnp, n_steps, u_stretch, partner, u_accept, a)
            except _lib.RoboBadShape:   # the LIBRARY declined (half an ensemble exceeds the batch workspace) -> host sampler;
                # Python-side shape asserts of DeviceGP.mcmc_run are bugs and propagate.  (A NaN / inf proposal is
                # out of bounds on the device, i.e. -inf and rejected; emcee 2 on the host raises "lnprob returned NaN"
                # for it -- neither can be produced from finite walkers by a stretch move)
                logger.info("device chain declined (%s); host sampler", _lib.last_error())
                return None
        return run

    def loglikelihood_batch(self, thetas):
        """log p(y | X, theta) + log prior for a batch (k, P); out-of-bounds / non-PD -> -inf
        (gaussian_process_mcmc.py:185-202)."""
        thetas = np.atleast_2d(np.asarray(thetas, dtype=np.float64))
        out = np.full(thetas.shape[0], -np.inf)
        ok = ~np.any((-20 > thetas) + (thetas > 20), axis=1) & np.all(np.isfinite(thetas), axis=1)
        if np.any(ok):
            if self._walker_shard() and self.walker_gps:
                ll, st = self._multi().loglik_batch([self.gp] + self.walker_gps, thetas[ok], self.mean)
            else:
                ll, st = self.gp.loglik_batch(thetas[ok], self.mean)
            ll = np.where(st == _lib.OK, ll, -np.inf)
            if self.prior is not None:
                if hasattr(self.prior, "lnprob_batch"):
                    ll = ll + self.prior.lnprob_batch(thetas[ok])
                else:
                    ll = ll + np.array([self.prior.lnprob(t) for t in thetas[ok]])
            out[ok] = ll
        return out

    def loglikelihood(self, theta):
        return float(self.loglikelihood_batch(np.asarray(theta)[None, :])[0])

    # ---- posterior ------------------------------------------------------------------------------
    @BaseModel._check_shapes_predict
    def predict(self, X_test, **kwargs):
        if not self.is_trained:
            raise Exception('Model has to be trained first!')
        if self.sample_shard:
            from robo_amd import sharding
            _, rank, world = sharding.dist_info()
            if world > 1:
                # mixture from per-rank partial sums in TWO exchanges, like np.var's two passes
                # (gaussian_process_mcmc.py:239): first (sum mu_s, sum var_s) -> the mixture mean m, then
                # sum (mu_s - m)^2.  The one-pass form mean(mu^2) - mean(mu)^2 cancels catastrophically when
                # |mu| is large against the spread of the mu_s (normalize_output=False).
                b, e = sharding.shard_range(len(self.models), rank, world)
                S = len(self.models)
                mus = []
                part = np.zeros((2, X_test.shape[0]))
                for model in self.models[b:e]:
                    mu_s, var_s = model.predict(X_test)
                    mus.append(mu_s)
                    part += np.stack((mu_s, var_s))
                tot = sharding.allgather_ordered_sum(part) / S
                m = tot[0]
                dev2 = np.zeros(X_test.shape[0])
                for mu_s in mus:
                    dev2 += (mu_s - m) ** 2
                v = sharding.allgather_ordered_sum(dev2) / S + tot[1]
                return m, np.clip(v, np.finfo(v.dtype).eps, np.inf)
        gps = [getattr(m, "gp", None) for m in self.models]
        if self.devices and len(self.models) >= 1 and all(isinstance(g, _lib.DeviceGP) for g in gps) and \
                all(m.is_trained for m in self.models) and self._on_their_slots():
            # the samples live on several devices: per-device posteriors, gathered and mixed on the first device
            groups = self._groups()
            m0 = self.models[0]
            Xn = m0.normalize(X_test) if hasattr(m0, "normalize") else m0._normalised(X_test)
            ctxs = self._multi().ctxs
            cands = [_lib.Candidates(ctxs[g], Xn) if (groups[g] or g == 0) else None for g in range(len(ctxs))]
            try:
                return self._multi().predict_mixture([[m.gp for m in grp] for grp in groups], cands)
            finally:
                for c in cands:
                    if c is not None:
                        c.close()
        if all(isinstance(g, _lib.DeviceGP) for g in gps) and all(m.is_trained for m in self.models):
            # all samples live on the device: S posteriors on one candidate upload + mixture kernel
            m0 = self.models[0]
            Xn = m0.normalize(X_test) if hasattr(m0, "normalize") else m0._normalised(X_test)
            cand = _lib.Candidates(gps[0].ctx, Xn)
            try:
                return _lib.predict_mixture(gps, cand)
            finally:
                cand.close()
        mu = np.zeros([len(self.models), X_test.shape[0]])
        var = np.zeros([len(self.models), X_test.shape[0]])
        for i, model in enumerate(self.models):
            mu[i], var[i] = model.predict(X_test)
        m = mu.mean(axis=0)
        # total variance of the mixture (Hutter et al.), gaussian_process_mcmc.py:235-239
        v = np.var(mu, axis=0) + np.mean(var, axis=0)
        v = np.clip(v, np.finfo(v.dtype).eps, np.inf)
        return m, v

    def get_incumbent(self):
        inc, inc_value = super(GaussianProcessMCMC, self).get_incumbent()
        if self.normalize_input:
            inc = normalization.zero_one_unnormalization(inc, self.lower, self.upper)
        if self.normalize_output:
            inc_value = normalization.zero_mean_unit_var_unnormalization(inc_value, self.y_mean, self.y_std)
        return inc, inc_value

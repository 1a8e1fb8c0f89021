"""Entropy search (Hennig & Schuler 2012): information gain about the location of the minimum.

Constructor, attributes (``zb``, ``lmb``, ``logP``, ``dlogPdMu``, ``dlogPdSigma``,
``dlogPdMudMu``, ``W``, ``sn2``) and the update/compute protocol of
robo/acquisition_functions/information_gain.py:19-272:

  update(model)   sample Nb representer points with an ensemble sampler on the proposal
                  acquisition (LogEI by default; 50 steps, up to 5 restarts while any log-value
                  is infinite, :132-151), posterior over them with full covariance, EP for
                  log p_min and its derivatives (host, robo_amd/util/epmgp.py), outcome quantiles W.
  compute(X)      for every candidate the innovation of the belief at the representer points and
                  the resulting expected entropy change -- the reference's Python loop with one
                  (Nb+1)-point covariance solve per candidate (:112-116, :253-272) is one batched
                  device call here (robo_ig_eval_cand: cross-covariances by fp64 MFMA GEMM over the
                  rows of L^-1 K*, quadratic forms by a second GEMM, entropy kernel).

Mirrored reference behaviour that changes numbers: ``v_ = v - sn2`` subtracts the noise from an
already noise-free predictive variance (:257-259); covariances between x and the representer points
come through ``predict(full_cov=True)`` and are therefore floored at eps, negative ones included
(gaussian_process.py:290-294); NaN / +inf gains become ``-sys.float_info.max`` (:119-120); out-of-box
candidates get ``np.spacing(1)`` (:215-218).  ``derivative=True`` (central differences, "Not tested!"
in the reference, :99) is not provided.
"""
import logging

import numpy as np

from robo_amd import _lib
from robo_amd.acquisition_functions.base_acquisition import BaseAcquisitionFunction
from robo_amd.acquisition_functions.log_ei import LogEI
from robo_amd.util import epmgp
from robo_amd.util.ensemble_sampler import EnsembleSampler

logger = logging.getLogger(__name__)


def outcome_quantiles(Np):
    from scipy.stats import norm
    return norm.ppf(np.linspace(1. / (Np + 1), 1 - 1. / (Np + 1), Np))[np.newaxis, :]


class InformationGain(BaseAcquisitionFunction):

    def __init__(self, model, lower, upper, Nb=50, Np=400, sampling_acquisition=None,
                 sampling_acquisition_kw={"par": 0.0}, rng=None, **kwargs):
        self.Nb = Nb
        super(InformationGain, self).__init__(model)
        self.lower = lower
        self.upper = upper
        self.D = self.lower.shape[0]
        self.sn2 = None
        if sampling_acquisition is None:
            sampling_acquisition = LogEI
        self.sampling_acquisition = sampling_acquisition(model, **sampling_acquisition_kw)
        self.Np = Np
        self.rng = np.random.RandomState(np.random.randint(0, 10000)) if rng is None else rng
        self._ep = None
        # candidate shard over GPUs (explicit opt-in, like RandomSampling.shard): the representer points come from an
        # ensemble sampler whose stream -- as in the reference, where emcee ignores the RandomState object handed to it
        # (information_gain.py:139-142) -- is seeded from OS entropy, i.e. differs from rank to rank; with shard = True
        # update() adopts rank 0's points on every rank, so that all shards are scored against ONE p_min
        self.shard = False

    # ---- representer points ------------------------------------------------------------------
    def sampling_acquisition_wrapper(self, x):
        if np.any(x < self.lower) or np.any(x > self.upper):
            return -np.inf
        return self.sampling_acquisition(np.array([x]))[0]

    def _proposal_batch(self, X):
        """log proposal density for a batch of walkers: -inf outside the box, acquisition inside"""
        X = np.atleast_2d(X)
        out = np.full(X.shape[0], -np.inf)
        inside = ~(np.any(X < self.lower, axis=1) | np.any(X > self.upper, axis=1))
        if np.any(inside):
            out[inside] = np.asarray(self.sampling_acquisition(X[inside])).reshape(-1)
        return out

    def sample_representer_points(self):
        self.sampling_acquisition.update(self.model)
        for _ in range(5):
            restarts = self.lower + (self.upper - self.lower) * self.rng.uniform(size=(self.Nb, self.D))
            sampler = EnsembleSampler(self.Nb, self.D, lnprob_batch=self._proposal_batch)
            self.zb, self.lmb, _ = sampler.run_mcmc(restarts, 50, rstate0=self.rng)
            if not np.any(np.isinf(self.lmb)):
                break
            logger.debug("representer proposal hit -inf, resampling")
        if len(self.zb.shape) == 1:
            self.zb = self.zb[:, None]
        if len(self.lmb.shape) == 1:
            self.lmb = self.lmb[:, None]

    # ---- update / compute ------------------------------------------------------------------------
    def update(self, model):
        self.model = model
        self.sn2 = self.model.get_noise()
        self.sample_representer_points()
        if self.shard:
            from robo_amd import sharding
            if sharding.dist_info()[2] > 1:
                zb, lmb = np.asarray(self.zb, dtype=np.float64), np.asarray(self.lmb, dtype=np.float64)
                row0 = sharding.allgather_rows(np.concatenate([zb.ravel(), lmb.ravel()]))[0]
                self.zb, self.lmb = row0[:zb.size].reshape(zb.shape), row0[zb.size:].reshape(lmb.shape)
        mu, var = self.model.predict(np.array(self.zb), full_cov=True)
        self.logP, self.dlogPdMu, self.dlogPdSigma, self.dlogPdMudMu = epmgp.joint_min(mu, var,
                                                                                       with_derivatives=True)
        self.W = outcome_quantiles(self.Np)
        self.logP = np.reshape(self.logP, (self.logP.shape[0], 1))
        self._ep = _lib.EPState(self.logP, self.lmb, self.W, self.dlogPdMu, self.dlogPdSigma, self.dlogPdMudMu)

    def _native(self):
        return isinstance(getattr(self.model, "gp", None), _lib.DeviceGP) and self.model.is_trained

    def _gains(self, X_test, want_values=True):
        if not (np.all(np.isfinite(self.lmb))):
            raise ValueError("lmb should not be infinite.")
        if self._native() and getattr(self.model, "devices", None) and \
                X_test.shape[0] >= len(self.model.devices):
            return self._gains_multi(X_test, want_values)
        if self._native():
            model = self.model
            model._materialise()
            norm = model.normalize if hasattr(model, "normalize") else model._normalised
            ctx = model.gp.ctx
            cand = _lib.Candidates(ctx, norm(X_test))
            rep = _lib.Candidates(ctx, norm(np.array(self.zb)))
            try:
                return _lib.ig_eval(model.gp, cand, rep, self._ep, self.sn2, want_values)
            finally:
                cand.close()
                rep.close()
        # any other model: innovations inputs from its own predict / predict_variance, entropy
        # algebra still on the device
        v = np.asarray(self.model.predict(X_test)[1], dtype=np.float64).reshape(-1)
        s = np.array([np.asarray(self.model.predict_variance(np.array(self.zb), x[None, :])).reshape(-1)
                      for x in X_test])
        vals = _lib.ig_from_moments(_lib.default_context(), s, v, self._ep, self.sn2)
        am = int(np.argmax(vals))
        return vals, vals[am], am

    def _gains_multi(self, X_test, want_values):
        """candidate shard over the devices of ONE process (``GaussianProcess(devices=...)``): every device scores its
        contiguous slice against its replica of the model and the SAME representer points / EP state
        (robo_ig_eval_cand_multi); the per-device (max, index) are reduced with np.argmax's tie-break (values are per
        candidate: sharding changes none)"""
        model = self.model
        model._materialise()
        norm = model.normalize if hasattr(model, "normalize") else model._normalised
        multi = model._multi()
        shards = _lib.CandidateShards.split(multi.ctxs, norm(X_test))
        zbn = norm(np.array(self.zb))
        reps = [_lib.Candidates(c, zbn) for c in multi.ctxs]
        try:
            vals, mx, am, _ = multi.ig(model._all_gps(), shards, reps, self._ep, self.sn2, want_values)
        finally:
            for h in reps:
                h.close()
            shards.close()
        return vals, mx, am

    # ---- the reference's per-candidate building blocks, kept callable ------------------------------------------
    # compute() never goes through them (one batched device call scores all candidates); they exist because they are
    # public methods of the reference class (information_gain.py:60-66, :199-252, :253-272) and return what those return.
    def loss_function(self, logP, lmb, lPred, *args):
        """entropy change of the belief over the representer points for each predicted outcome (columns of lPred),
        relative to the current belief logP, both measured against the proposal measure lmb -> array (1, Np)"""
        p_now, p_new = np.exp(logP), np.exp(lPred)
        entropy_now = -np.sum(p_now * (logP + lmb))
        return np.array([-np.sum(p_new * (lPred + lmb), axis=0) - entropy_now])

    def innovations(self, x, rep):
        """change of the posterior at the representer points ``rep`` (Nb, D) if ``x`` (1, D) were evaluated:
        (stochastic innovation of the mean (Nb, 1), deterministic innovation of the covariance (Nb, Nb)); the quirks of
        :257-259 included (the noise is subtracted from an already noise-free predictive variance)"""
        v = np.asarray(self.model.predict(x)[1], dtype=np.float64).reshape(-1, 1)
        v_less = v - self.sn2
        cross = np.asarray(self.model.predict_variance(rep, x), dtype=np.float64)
        scaled = cross.dot(np.linalg.inv(v_less))
        return scaled.dot(np.linalg.cholesky(v + 1e-10)), -scaled.dot(cross.T)

    def dh_fun(self, x, derivative=False):
        """information gain of ONE candidate x (1, D) -> array([dH]) (:199-252; np.spacing(1) outside the box)"""
        if derivative:
            raise NotImplementedError("InformationGain.dh_fun: derivative=True (central differences, untested in the "
                                      "reference) is not provided")
        x = np.atleast_2d(np.asarray(x, dtype=np.float64))
        if not np.all(np.isfinite(self.lmb)):
            raise ValueError("lmb should not be infinite.")
        if np.any(x < self.lower) or np.any(x > self.upper):
            return np.array([[np.spacing(1)]]), np.array([[np.zeros((x.shape[1], 1))]])
        return np.array([self._gains(x[:1])[0][0]])

    def compute(self, X_test, derivative=False, **kwargs):
        if derivative:
            raise NotImplementedError("InformationGain: derivative=True (central differences, untested in the "
                                      "reference) is not provided")
        acq, _, _ = self._gains(X_test)
        outside = np.any(X_test < self.lower, axis=1) | np.any(X_test > self.upper, axis=1)
        acq[outside] = np.spacing(1)
        return acq

    def argmax(self, X_test):
        outside = np.any(X_test < self.lower, axis=1) | np.any(X_test > self.upper, axis=1)
        if np.any(outside):
            return int(np.argmax(self.compute(X_test)))
        return int(self._gains(X_test, want_values=False)[2])

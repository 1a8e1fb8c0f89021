"""Acquisition marginalised over GP hyper-parameter samples.

Semantics of robo/acquisition_functions/marginalization.py:11-121: one estimator (a deep
copy of the wrapped acquisition function) per ``model.models[i]``; ``update`` re-points each
estimator at the retrained sub-model; ``compute`` returns the mean over samples, accumulated
in sample order (NumPy's axis-0 mean).

When every sub-model is a device-resident robo_amd GaussianProcess and the wrapped function
is EI/LogEI/PI/LCB, the S posteriors, S acquisition vectors, their ordered sum and the final
argmax are produced by ONE C-ABI call on a shared candidate upload
(robo_acq_eval_marginal_cand); otherwise the estimators are evaluated one by one exactly
like the reference does.
"""
import logging
from copy import deepcopy

import numpy as np

from robo_amd import _lib
from robo_amd.acquisition_functions.base_acquisition import BaseAcquisitionFunction, ClosedFormAcquisition

logger = logging.getLogger(__name__)


class MarginalizationGPMCMC(BaseAcquisitionFunction):

    def __init__(self, acquisition_func):
        self.acquisition_func = acquisition_func
        self.model = acquisition_func.model
        self.cost_model = getattr(acquisition_func, "cost_model", None)
        self.estimators = []
        self._build_estimators()
        self.last_max = None
        self.last_argmax = None
        # sample shard (SURVEY.md 8e axis 2): with one process per GPU and ``sample_shard = True`` every rank
        # evaluates only ITS hyper-parameter samples on all candidates; the per-rank partial sums are exchanged
        # and added in rank order (robo_amd.sharding.allgather_ordered_sum)
        self.sample_shard = False

    def _build_estimators(self):
        for i in range(len(self.model.models)):
            # detach the (possibly device-resident) model while copying: the copy gets its own
            # sub-model right below, as in marginalization.py:36-40
            model, self.acquisition_func.model = self.acquisition_func.model, None
            try:
                estimator = deepcopy(self.acquisition_func)
            finally:
                self.acquisition_func.model = model
            estimator.model = self.model.models[i]
            if self.cost_model is not None and len(self.cost_model.models) > 0:
                estimator.cost_model = self.cost_model.models[i]
            self.estimators.append(estimator)

    def update(self, model, cost_model=None, **kwargs):
        self.model = model
        if cost_model is not None:
            self.cost_model = cost_model
        if len(self.estimators) != len(self.model.models):
            self.estimators = []
            self._build_estimators()
        for i in range(len(self.model.models)):
            if cost_model is not None:
                self.estimators[i].update(self.model.models[i], self.cost_model.models[i], **kwargs)
            else:
                self.estimators[i].update(self.model.models[i], **kwargs)

    def _device_groups(self):
        """Single-process multi-GPU (``GaussianProcessMCMC(devices=...)``): the estimators grouped by the device slot
        their sub-model lives on, or None when this is not such a model / not a fused closed-form case."""
        devices = getattr(self.model, "devices", None)
        if not devices or self._shard() is not None or not self.estimators:
            return None
        if not isinstance(self.acquisition_func, ClosedFormAcquisition):
            return None
        multi = _lib.multi_for(devices)
        groups = [[] for _ in multi.ctxs]
        for e in self.estimators:
            gp = getattr(e.model, "gp", None)
            if not isinstance(gp, _lib.DeviceGP) or not getattr(e.model, "is_trained", False):
                return None
            slot = next((g for g, c in enumerate(multi.ctxs) if c is gp.ctx), None)
            if slot is None or (groups[slot + 1:] and any(groups[slot + 1:])):
                return None                          # not on the list, or not in contiguous sample order
            groups[slot].append(e)
        return multi, groups

    def _multi_eval(self, X_test, want_values):
        """sample shard over the devices of ONE process (robo_acq_eval_marginal_cand_multi): every device accumulates its
        samples' acquisition values on all candidates; the partial sums are added in device order on the first device"""
        multi, groups = self._device_groups()
        if isinstance(X_test, _lib.Candidates):
            X_test = self._host_points(X_test)       # every device needs the whole batch: through the host, once
        m0 = self.estimators[0].model
        Xn = (m0.normalize if hasattr(m0, "normalize") else m0._normalised)(X_test)
        cands = [_lib.Candidates(c, Xn) if (groups[g] or g == 0) else None for g, c in enumerate(multi.ctxs)]
        ref = self.estimators[0]
        try:
            vals, mx, am, flags = multi.acq_marginal([[e.model.gp for e in grp] for grp in groups], ref.kind, ref.par,
                                                     [[e._eta(None) for e in grp] for grp in groups], cands, want_values)
        finally:
            for c in cands:
                if c is not None:
                    c.close()
        self.last_max, self.last_argmax = mx, am
        return vals, flags

    def _native(self):
        if self._shard() is not None:
            return False
        if not isinstance(self.acquisition_func, ClosedFormAcquisition) or not self.estimators:
            return False
        gps = [getattr(e.model, "gp", None) for e in self.estimators]
        return all(isinstance(g, _lib.DeviceGP) for g in gps) and len({id(g.ctx) for g in gps}) == 1 \
            and all(getattr(e.model, "is_trained", False) for e in self.estimators)

    def _shard(self):
        # the opt-in flag FIRST: dist_info() is collective on first use (the communicator id is broadcast, every rank
        # joins ncclCommInitRank) and must not run because a torch process group merely exists
        if not self.sample_shard:
            return None
        from robo_amd import sharding
        _, rank, world = sharding.dist_info()
        if world == 1:
            return None
        return sharding.shard_range(len(self.estimators), rank, world)

    def _sharded_eval(self, X_test):
        """mean over ALL samples from per-rank partial sums; -> values (M,), identical on every rank"""
        from robo_amd import sharding
        b, e = self._shard()
        est = self.estimators[b:e]
        S = len(self.estimators)
        if isinstance(X_test, _lib.Candidates):
            # a device-generated batch lives in the normalised space of ITS maximiser's model; the sample shard
            # evaluates host coordinates (every rank must see the same points): bring them back once
            X_test = self._host_points(X_test)
        comm = sharding.comm()
        # which exchange runs must not depend on the rank: decided from the CLASSES of all estimators' models
        fused = isinstance(self.acquisition_func, ClosedFormAcquisition) and \
            all(hasattr(x.model, "acquisition") and hasattr(x.model, "gp") for x in self.estimators)
        if fused:
            # partial sums on the device, all-gathered and added in rank order inside the library
            # (robo_acq_eval_marginal_cand_sharded); a rank without samples (S < world) takes part with an empty sum
            for x in est:
                if not x.model.is_trained:
                    raise Exception('Model has to be trained first!')
                x.model._materialise()
            ref = est[0] if est else self.estimators[0]
            if est:
                m0 = ref.model
                Xn = (m0.normalize if hasattr(m0, "normalize") else m0._normalised)(X_test)
            else:
                Xn = np.zeros_like(np.asarray(X_test, dtype=np.float64))      # never evaluated: no local sample
            cand = _lib.Candidates(comm.ctx, Xn)
            try:
                vals, mx, am, flags = comm.acq_marginal_sharded([x.model.gp for x in est], S, ref.kind, ref.par,
                                                                [x._eta(None) for x in est], cand)
            finally:
                cand.close()
            self.last_max, self.last_argmax = mx, am
            if ref.kind == "ei" and flags & _lib.FLAG_NEGATIVE_EI:
                raise ValueError                      # the flags are OR-ed over all ranks: every rank raises
            return vals
        part = np.zeros(X_test.shape[0])
        for x in est:
            part = part + np.asarray(x.compute(X_test), dtype=np.float64).reshape(-1)
        return sharding.allgather_ordered_sum(part) / S

    def _host_points(self, cand):
        """coordinates of a device candidate batch in the caller's input space"""
        m0 = self.estimators[0].model
        P = cand.points()
        lower, upper = np.asarray(m0.lower, dtype=np.float64), np.asarray(m0.upper, dtype=np.float64)
        if hasattr(m0, "normalize") and not getattr(m0, "normalize_input", False):
            raise TypeError("device-generated candidates are not supported with Fabolas sub-models")
        return lower + (upper - lower) * P

    def _native_eval(self, X_test, want_values):
        est = self.estimators
        models = [e.model for e in est]
        gps = [m.gp for m in models]
        # every estimator asks its own sub-model for the incumbent (marginalization.py:40, ei.py:68)
        eta = np.array([e._eta(None) for e in est])
        # FabolasGP sub-models map their inputs through normalize() ([0,1] scaling of the configuration
        # columns + basis function on the fidelity column, fabolas_gp.py:122-126); plain GPs through the
        # [0,1] normalisation
        norm = models[0].normalize if hasattr(models[0], "normalize") else models[0]._normalised
        cand = X_test if isinstance(X_test, _lib.Candidates) else _lib.Candidates(gps[0].ctx, norm(X_test))
        try:
            vals, mx, am, flags = _lib.acq_marginal(gps, est[0].kind, est[0].par, eta, cand, want_values)
        finally:
            if cand is not X_test:
                cand.close()
        self.last_max, self.last_argmax = mx, am
        return vals, flags

    def compute(self, X_test, derivative=False):
        if not derivative and self._shard() is not None:
            return self._sharded_eval(X_test)
        fused = None
        if not derivative and self._device_groups() is not None:
            fused = self._multi_eval
        elif not derivative and self._native():
            fused = self._native_eval
        if fused is not None:
            vals, flags = fused(X_test, True)
            if self.estimators[0].kind == "ei":
                if flags & _lib.FLAG_ZERO_SIGMA:
                    # some estimator would have returned [[0]] (ei.py:72-74); keep the reference's
                    # per-estimator behaviour by falling through to the one-by-one evaluation
                    pass
                elif flags & _lib.FLAG_NEGATIVE_EI:
                    raise ValueError
                else:
                    return vals
            else:
                return vals
        if isinstance(X_test, _lib.Candidates):
            X_test = self._host_points(X_test)       # host loop below: the reference's per-estimator evaluation
        acquisition_values = np.zeros([len(self.model.models), X_test.shape[0]])
        by_ctx = {}
        for i, e in enumerate(self.estimators):
            # every context an estimator drives: its model's and, for the per-unit-cost form, its cost model's -- a context
            # (stream, pinned read-back, scratch) is not thread-safe, so it may belong to ONE thread only
            key = []
            for mdl in (getattr(e, "model", None), getattr(e, "cost_model", None)):
                gp = getattr(mdl, "gp", None)
                if mdl is not None:
                    key.append(id(gp.ctx) if isinstance(gp, _lib.DeviceGP) else None)
            by_ctx.setdefault(tuple(key), []).append(i)
        seen = [c for key in by_ctx for c in set(key)]
        disjoint = len(seen) == len(set(seen)) and None not in seen
        if len(by_ctx) > 1 and disjoint and getattr(self.model, "devices", None):
            # sub-models on several devices of this process (any acquisition function, e.g. the information gain per unit
            # cost of Fabolas): one host thread per device walks its estimators -- the library calls release the GIL, so
            # the devices work at the same time; the mean is still taken in sample order
            from concurrent.futures import ThreadPoolExecutor

            def run(idx):
                for i in idx:
                    acquisition_values[i] = self.estimators[i].compute(X_test, derivative=derivative)
            with ThreadPoolExecutor(max_workers=len(by_ctx)) as pool:
                for f in [pool.submit(run, idx) for idx in by_ctx.values()]:
                    f.result()
            return acquisition_values.mean(axis=0)
        for i in range(len(self.model.models)):
            acquisition_values[i] = self.estimators[i].compute(X_test, derivative=derivative)
        return acquisition_values.mean(axis=0)

    def argmax(self, X_test):
        if self._shard() is not None:
            return int(np.argmax(self._sharded_eval(X_test)))
        fused = self._multi_eval if self._device_groups() is not None else (self._native_eval if self._native() else None)
        if fused is not None:
            _, flags = fused(X_test, False)
            if self.estimators[0].kind != "ei" or not flags & (_lib.FLAG_ZERO_SIGMA | _lib.FLAG_NEGATIVE_EI):
                return int(self.last_argmax)
            # an estimator would have collapsed to [[0]] / raised (ei.py:72-74,86-88): same path as compute()
        return int(np.argmax(self.compute(X_test)))

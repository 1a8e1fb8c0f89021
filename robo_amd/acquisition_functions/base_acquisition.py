"""Acquisition plugin surface -- the contract of
robo/acquisition_functions/base_acquisition.py:4-69: ``__init__(model)``, ``update(model)``,
``compute(X, derivative=False, **kw)``, ``__call__ = compute``, ``get_json_data()``.

:class:`ClosedFormAcquisition` is the shared host shim of EI / LogEI / PI / LCB.  All
arithmetic runs on the device:

* model is a robo_amd GaussianProcess  -> fused path: cross-gram, triangular solve,
  variance/mean, acquisition and argmax in one C-ABI call (robo_acq_eval);
* any other BaseModel plugin            -> ``model.predict`` supplies (mean, var) and only the
  element-wise kernel runs (robo_acq_eval_moments).
"""
import abc
import logging

import numpy as np

from robo_amd import _lib

logger = logging.getLogger(__name__)


class BaseAcquisitionFunction(object):
    __metaclass__ = abc.ABCMeta

    def __init__(self, model):
        self.model = model

    def update(self, model):
        """Called by the solver after the model was retrained."""
        self.model = model

    @abc.abstractmethod
    def compute(self, x, derivative=False):
        """Acquisition values at x (N, D) -> (N,)."""

    def __call__(self, x, **kwargs):
        return self.compute(x, **kwargs)

    def get_json_data(self):
        return {"type": __name__}


class ClosedFormAcquisition(BaseAcquisitionFunction):
    """EI / LogEI / PI / LCB on (mean, var, eta); subclasses set ``kind`` and the guards."""

    kind = None
    needs_eta = True

    def __init__(self, model, par=0.0, **kwargs):
        super(ClosedFormAcquisition, self).__init__(model)
        self.par = par
        self.last_max = None
        self.last_argmax = None

    def _is_native(self):
        return hasattr(self.model, "acquisition") and hasattr(self.model, "gp")

    def _eta(self, eta):
        if not self.needs_eta:
            return 0.0
        if eta is None:
            _, eta = self.model.get_incumbent()
        return float(eta)

    def _evaluate(self, X, eta):
        """-> (values (N,), flags)"""
        eta = self._eta(eta)
        if self._is_native():
            vals, mx, am, flags = self.model.acquisition(self.kind, self.par, eta, X)
        else:
            m, v = self.model.predict(X)
            vals, mx, am, flags = _lib.acq_from_moments(_lib.default_context(), self.kind, self.par, eta,
                                                        np.asarray(m, dtype=np.float64).ravel(),
                                                        np.asarray(v, dtype=np.float64).ravel())
        self.last_max, self.last_argmax = mx, am
        return vals, flags

    def argmax(self, X, eta=None):
        """Index of the best candidate without copying the values back (large-M maximisers)."""
        eta = self._eta(eta)
        if self._is_native():
            _, mx, am, flags = self.model.acquisition(self.kind, self.par, eta, X, want_values=False)
            self.last_max, self.last_argmax = mx, am
            if self.kind == "ei":
                # the reference's guards apply when maximising too (ei.py:72-74,86-88): any zero sigma
                # collapses the batch to [[0]] (argmax 0), any negative EI raises
                if flags & _lib.FLAG_ZERO_SIGMA:
                    return 0
                if flags & _lib.FLAG_NEGATIVE_EI:
                    raise ValueError
            return int(am)
        return int(np.argmax(self.compute(X, eta=eta)))

    def argmax_sharded(self, comm, X_slice, global_offset):
        """Candidate shard (robo_amd.sharding.sharded_argmax): this rank's slice of the candidate matrix, whose first
        row has global index ``global_offset`` -> the GLOBAL np.argmax index, identical on every rank.  Posterior,
        acquisition, local argmax, the RCCL all-gather of the per-rank incumbents and the cross-rank tie-break are one
        library call (robo_acq_eval_cand_sharded); the reference's EI guards act on the flags OR-ed over all ranks,
        i.e. exactly as they would on the unsharded batch."""
        eta = self._eta(None)
        n_here = int(np.asarray(X_slice).shape[0])
        if not self._is_native():
            from robo_amd import sharding
            if n_here == 0:                           # more ranks than candidates: an empty shard still joins the exchange
                return sharding.allgather_argmax(-np.inf, -1)[1]
            vals = np.asarray(self.compute(X_slice), dtype=np.float64).reshape(-1)
            if vals.shape[0] != X_slice.shape[0]:
                vals = np.zeros(X_slice.shape[0])
            j = int(np.argmax(vals))
            return sharding.allgather_argmax(float(vals[j]), global_offset + j)[1]
        model = self.model
        if not model.is_trained:
            raise Exception('Model has to be trained first!')
        if n_here == 0:
            # an empty shard takes part in the SAME collective the other ranks issue inside robo_acq_eval_cand_sharded
            from robo_amd import sharding
            mx, am, flags = sharding.exchange_best(comm, lambda: None)
        else:
            model._materialise()
            norm = model.normalize if hasattr(model, "normalize") else model._normalised
            cand = _lib.Candidates(model.gp.ctx, norm(X_slice))
            try:
                _, mx, am, _, flags = comm.acq_sharded(model.gp, self.kind, self.par, eta, cand, global_offset)
            finally:
                cand.close()
        self.last_max, self.last_argmax = mx, am
        if self.kind == "ei":
            if flags & _lib.FLAG_ZERO_SIGMA:
                return 0
            if flags & _lib.FLAG_NEGATIVE_EI:
                raise ValueError
        return int(am)

    def _moment_gradients(self, X):
        """(mean, var, d mean / d x (M, D), d var / d x (M, D)) from ``model.predictive_gradients`` -- the
        protocol of ei.py:80-85 / pi.py:65-71 / lcb.py:66-68 (GPy shapes: dmdx (M, D, 1), dvdx (M, D))."""
        if not hasattr(self.model, "predictive_gradients"):
            raise NotImplementedError("%s: derivative=True needs model.predictive_gradients"
                                      % self.__class__.__name__)
        m, v = self.model.predict(X)
        dmdx, dvdx = self.model.predictive_gradients(X)
        dmdx = np.asarray(dmdx, dtype=np.float64)
        if dmdx.ndim == 3:
            dmdx = dmdx[:, :, 0]
        return np.asarray(m, dtype=np.float64), np.asarray(v, dtype=np.float64), dmdx, \
            np.asarray(dvdx, dtype=np.float64)

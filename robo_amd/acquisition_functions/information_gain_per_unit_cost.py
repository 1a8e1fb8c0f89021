"""Information gain per unit cost (Swersky et al. 2013; Fabolas' acquisition).

Semantics of robo/acquisition_functions/information_gain_per_unit_cost.py:8-154:
``compute(X) = dH(X) / (exp(cost_model.predict(X).mean) + overhead)``; representer points are
sampled over the NON-environmental dimensions only, with the proposal acquisition evaluated at
the point projected to the upper bound of the environmental ones, and are then extended by one
column per environmental dimension.  The reference fills that column with the COUNT of
environmental dimensions instead of their upper bound (:151-153); identical for Fabolas (one
fidelity dimension, upper bound 1) -- mirrored as written.
"""
import numpy as np

from robo_amd.acquisition_functions.information_gain import InformationGain
from robo_amd.util.ensemble_sampler import EnsembleSampler


class InformationGainPerUnitCost(InformationGain):

    def __init__(self, model, cost_model, lower, upper, is_env_variable, sampling_acquisition=None,
                 n_representer=50, Np=400, rng=None):
        self.cost_model = cost_model
        self.n_dims = lower.shape[0]
        self.is_env = is_env_variable
        self.overhead = 0
        super(InformationGainPerUnitCost, self).__init__(model, lower, upper, Nb=n_representer, Np=Np,
                                                         sampling_acquisition=sampling_acquisition, rng=rng)

    def update(self, model, cost_model, overhead=None):
        self.cost_model = cost_model
        self.overhead = 0 if overhead is None else overhead
        super(InformationGainPerUnitCost, self).update(model)

    def compute(self, X, derivative=False):
        if len(X.shape) == 1:
            X = X[np.newaxis, :]
        if derivative:
            raise NotImplementedError("Not implemented")
        log_cost = self.cost_model.predict(X)[0]
        dh = super(InformationGainPerUnitCost, self).compute(X, derivative=False)
        return dh / (np.exp(log_cost) + self.overhead)

    def argmax(self, X):
        return int(np.argmax(self.compute(X)))

    # ---- representer points live on the configuration sub-space ------------------------------------
    def _config_box(self):
        free = np.where(self.is_env == 0)
        return self.lower[free], self.upper[free]

    def _proposal_batch(self, X):
        X = np.atleast_2d(X)
        lower, upper = self._config_box()
        out = np.full(X.shape[0], -np.inf)
        inside = ~(np.any(X < lower, axis=1) | np.any(X > upper, axis=1))
        if np.any(inside):
            env = np.tile(self.upper[self.is_env == 1], (int(inside.sum()), 1))
            out[inside] = np.asarray(self.sampling_acquisition(np.concatenate((X[inside], env), axis=1))).reshape(-1)
        return out

    def sampling_acquisition_wrapper(self, x):
        return float(self._proposal_batch(np.asarray(x)[None, :])[0])

    def sample_representer_points(self):
        lower, upper = self._config_box()
        D = lower.shape[0]
        self.sampling_acquisition.update(self.model)
        for _ in range(5):
            restarts = np.random.uniform(low=lower, high=upper, size=(self.Nb, D))   # global RNG, like :131
            sampler = EnsembleSampler(self.Nb, D, lnprob_batch=self._proposal_batch)
            self.zb, self.lmb, _ = sampler.run_mcmc(restarts, 50, rstate0=self.rng)
            if not np.any(np.isinf(self.lmb)):
                break
        if np.any(np.isinf(self.lmb)):
            raise ValueError("Could not sample valid representer points! LogEI is -infinity")
        if len(self.zb.shape) == 1:
            self.zb = self.zb[:, None]
        if len(self.lmb.shape) == 1:
            self.lmb = self.lmb[:, None]
        n_env = self.upper[self.is_env == 1].shape[0]
        proj = np.ones([self.zb.shape[0], n_env]) * n_env      # sic (:151-153)
        self.zb = np.concatenate((self.zb, proj), axis=1)

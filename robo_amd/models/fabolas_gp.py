"""Fabolas GP models (dataset-size fidelity in the last input column) on the MI355X hot path.

Semantics of robo/models/fabolas_gp.py:12-164: the first D columns are [0,1]-normalised with
the given bounds, the last column s is passed through ``basis_func`` ((1-s)^2 for the loss
model, s for the cost model, robo/fmin/fabolas.py:130-131) BEFORE the kernel sees it
(:122-126); the incumbent is the training configuration with the lowest PREDICTED mean when
projected to the full data set s = 1 (:141-164), not the lowest observation.
The kernel is :class:`robo_amd.kernels.FabolasKernel` (device kind ROBO_KERNEL_FABOLAS).
"""
import numpy as np

from robo_amd.models.gaussian_process import GaussianProcess
from robo_amd.models.gaussian_process_mcmc import GaussianProcessMCMC
from robo_amd.util import normalization


def _fabolas_normalize(X, lower, upper, basis):
    Xn, _, _ = normalization.zero_one_normalization(X[:, :-1], lower, upper)
    return np.concatenate((Xn, basis(X[:, -1])[:, None]), axis=1)


class FabolasGP(GaussianProcess):

    def __init__(self, kernel, basis_function, prior=None, noise=1e-3, use_gradients=False,
                 normalize_output=False, lower=None, upper=None, rng=None, device=None, devices=None):
        self.basis_function = basis_function
        super(FabolasGP, self).__init__(kernel=kernel, prior=prior, noise=noise, use_gradients=use_gradients,
                                        normalize_output=normalize_output, normalize_input=False, lower=lower,
                                        upper=upper, rng=rng, device=device, devices=devices)

    def normalize(self, X):
        return _fabolas_normalize(X, self.lower, self.upper, self.basis_function)

    def train(self, X, y, do_optimize=True):
        self.original_X = X
        return super(FabolasGP, self).train(self.normalize(X), y, do_optimize)

    def _host_train_raw(self, X, y, alloc=True):
        self.original_X = X
        return self._host_train(self.normalize(X), y, alloc)

    def predict(self, X_test, full_cov=False, **kwargs):
        return super(FabolasGP, self).predict(self.normalize(X_test), full_cov)

    def sample_functions(self, X_test, n_funcs=1):
        return super(FabolasGP, self).sample_functions(self.normalize(X_test), n_funcs)

    def predictive_gradients(self, X_test):
        """as GaussianProcess.predictive_gradients, through normalize(): configuration columns scaled by the
        bounds, the fidelity column by the derivative of the basis function (central difference of the user's
        callable, h = 1e-6)"""
        X_test = np.asarray(X_test, dtype=np.float64)
        dm, dv = super(FabolasGP, self).predictive_gradients(self.normalize(X_test))
        s = X_test[:, -1]
        h = 1e-6
        scale = np.ones_like(X_test)
        scale[:, :-1] = 1.0 / (np.asarray(self.upper, dtype=np.float64) - np.asarray(self.lower, dtype=np.float64))
        scale[:, -1] = (self.basis_function(s + h) - self.basis_function(s - h)) / (2 * h)
        return dm * scale[:, :, np.newaxis], dv * scale

    def acquisition(self, kind, par, eta, X_test, want_values=True):
        from robo_amd import _lib
        if not isinstance(X_test, (_lib.Candidates, _lib.CandidateShards)):    # device batches are in the model's space already
            X_test = self.normalize(X_test)
        return super(FabolasGP, self).acquisition(kind, par, eta, X_test, want_values)

    def get_incumbent(self):
        """(configuration projected to s = 1, its predicted mean there)"""
        proj = np.concatenate((self.original_X[:, :-1], np.ones([self.original_X.shape[0], 1])), axis=1)
        # MIRRORED QUIRK: the reference normalises the projected points and then calls predict(), which
        # normalises them AGAIN (fabolas_gp.py:156-157): the configuration columns are scaled twice and the
        # basis function is applied to basis(1).  Kept as is so that the incumbent equals the reference's
        # (fixture tests/golden/ref_fabolas.npz, produced by the reference class).
        m, _ = self.predict(self.normalize(proj))
        best = np.argmin(m)
        return proj[best], m[best]


class FabolasGPMCMC(GaussianProcessMCMC):

    def __init__(self, kernel, basis_func, prior=None, n_hypers=20, chain_length=2000, burnin_steps=2000,
                 normalize_output=False, rng=None, lower=None, upper=None, noise=-8, device=None, devices=None):
        self.basis_func = basis_func
        self.hypers = None
        super(FabolasGPMCMC, self).__init__(kernel, prior, n_hypers, chain_length, burnin_steps,
                                            normalize_output=normalize_output, normalize_input=False, rng=rng,
                                            lower=lower, upper=upper, noise=noise, device=device, devices=devices)

    def _make_model(self, kernel, noise):
        return FabolasGP(kernel, basis_function=self.basis_func, normalize_output=self.normalize_output, noise=noise,
                         lower=self.lower, upper=self.upper, rng=self.rng, device=self.device)

    def _model_inputs(self, X):
        return _fabolas_normalize(X, self.lower, self.upper, self.basis_func)

    # get_incumbent is NOT overridden, as in the reference (fabolas_gp.py:12-102): it is
    # GaussianProcessMCMC's -- argmin of the observed y, returned in the model's own input space (configuration
    # columns in [0,1], basis-transformed fidelity column; normalize_input is False so nothing is mapped
    # back).  The Fabolas loop itself asks projected_incumbent_estimation (fmin/fabolas.py:254).

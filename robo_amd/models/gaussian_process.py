"""ML-II Gaussian process on the MI355X hot path.

Same constructor, attributes and method semantics as
robo/models/gaussian_process.py:14-352; george's role (gp.compute / log_likelihood /
predict) is played by the device GP behind the C ABI (include/robo_hip.h):

  train            :70-124   normalise, mean = mean(y), [optimise], fit (retry at noise*10)
  nll              :129-166  |theta|>20 -> 1e25; LinAlgError -> 1e25; non-finite -> 1e25
  optimize         :193-219  scipy L-BFGS-B on nll (finite differences, like the reference)
  predict          :251-296  un-normalise, variance floor eps (on device)
  predict_variance :221-248, sample_functions :298-332, get_incumbent :334-352
"""
import copy
import logging

import numpy as np
from scipy import optimize

from robo_amd import _lib
from robo_amd.models.base_model import BaseModel
from robo_amd.util import normalization

logger = logging.getLogger(__name__)


class GaussianProcess(BaseModel):

    def __init__(self, kernel, prior=None, noise=1e-3, use_gradients=False, normalize_output=False,
                 normalize_input=True, lower=None, upper=None, rng=None, device=None, devices=None):
        if rng is None:
            self.rng = np.random.RandomState(np.random.randint(0, 10000))
        else:
            self.rng = rng
        self.kernel = kernel
        self.gp = None                 # robo_amd._lib.DeviceGP (the reference holds a george.GP here)
        self.prior = prior
        self.noise = noise
        self.use_gradients = use_gradients
        self.normalize_output = normalize_output
        self.normalize_input = normalize_input
        self.X = None
        self.y = None
        self.hypers = []
        self.is_trained = False
        self.lower = lower
        self.upper = upper
        self.device = device
        # devices = [d0, d1, ...] (not in the reference, which is one process on one CPU): single-process multi-GPU.  The
        # final fit of train() is replicated on every listed device (deterministic: one replica per device, all fitted
        # concurrently) and the candidate batch of an acquisition maximisation is split over them
        # (robo_acq_eval_cand_multi); everything else -- hyper-parameter optimisation, predict() -- runs on devices[0].
        self.devices = _lib.resolve_devices(devices)
        self.replicas = []             # DeviceGPs on devices[1:], same data and factor as self.gp
        self._shard_cache = None       # per-device candidate handles of the last host batch (kept between maximisations)
        self._ctx_override = None      # GaussianProcessMCMC places its per-sample models on the contexts of its device list
        self._fitted_theta = None

    # ---- device handle management ----------------------------------------------------------
    def _multi(self):
        return _lib.multi_for(self.devices) if self.devices else None

    def _ctx(self):
        if self._ctx_override is not None:
            return self._ctx_override
        if self.devices:
            return self._multi().ctxs[0]
        return _lib.default_context(self.device)

    def _all_gps(self):
        return [self.gp] + list(self.replicas)

    def _ensure_gp(self, n, dim):
        if self.gp is None or self.gp.dim != dim or self.gp.n_max < n or self.gp.kind != self.kernel.kind:
            cap = max(127, int(n))
            if self.gp is not None and self.gp.n_max < n:
                cap = max(cap, 2 * self.gp.n_max)     # amortise growth over a BO run
            for g in self._all_gps():
                if g is not None:
                    g.close()
            head = self.kernel.fixed_head()      # (0.0,) for a george kernel without an amplitude factor, else ()
            self.gp = _lib.DeviceGP(self._ctx(), self.kernel.kind, cap, dim, fixed_head=head)
            self.replicas = [_lib.DeviceGP(c, self.kernel.kind, cap, dim, fixed_head=head) for c in self._multi().ctxs[1:]] \
                if self.devices else []
        return self.gp

    def _shards_for(self, Xn):
        """per-device candidate handles for a host batch: kept between calls of one batch size (a BO loop maximises over
        the same number of candidates every iteration) and only re-uploaded, like the single-device host-array path"""
        Xn = np.ascontiguousarray(Xn, dtype=np.float64)
        old = self._shard_cache
        if old is not None and old[0] == Xn.shape and old[0][0] <= 16384 * len(self.devices):
            shards = old[1]
            for g, c in enumerate(shards.shards):
                if c is not None:
                    c.set_points(Xn[shards.offsets[g]:shards.offsets[g] + c.m])
            return shards
        if old is not None:
            old[1].close()
        shards = _lib.CandidateShards.split(self._multi().ctxs, Xn)
        self._shard_cache = (Xn.shape, shards)
        return shards

    def _upload(self):
        """training data to the primary handle and, with a device list, to every replica"""
        if self.devices:
            self._multi().set_data(self._all_gps(), self.X, self.y)
        else:
            self.gp.set_data(self.X, self.y)

    def _fit_everywhere(self, theta):
        """the fit the model keeps: on every device of the list at once (robo_gp_fit_multi), else on the one handle"""
        if self.devices:
            return self._multi().fit(self._all_gps(), theta, self.mean)
        return self.gp.fit(theta, self.mean)

    def __deepcopy__(self, memo):
        # device memory is not copied: the copy re-fits lazily from its host state on first use
        # (MarginalizationGPMCMC deep-copies acquisition functions, marginalization.py:36,67)
        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            if k == "gp":
                new.gp = None
            elif k == "replicas":
                new.replicas = []
            elif k == "_shard_cache":
                new._shard_cache = None
            elif k == "_ctx_override":
                new._ctx_override = v             # a context is shared, not copied
            else:
                setattr(new, k, copy.deepcopy(v, memo))
        return new

    def __getstate__(self):
        d = dict(self.__dict__)
        d["gp"] = None
        d["replicas"] = []
        d["_shard_cache"] = None
        d["_ctx_override"] = None
        return d

    def _materialise(self):
        """(re)create the device state of a trained model that lost its handle (deepcopy/pickle)."""
        if self.gp is None and self.is_trained:
            self._ensure_gp(self.X.shape[0], self.X.shape[1])
            self._upload()
            self._set_transform()
            self._fit_everywhere(self._fitted_theta)

    def _device(self):
        """the device GP holding this model's data (re-created after deepcopy / unpickling)"""
        if self.gp is None:
            if self.X is None:
                raise Exception('Model has to be trained first!')
            self._materialise()
            if self.gp is None:                       # data set but never fitted (optimize() before train)
                self._ensure_gp(self.X.shape[0], self.X.shape[1])
                self._upload()
                self._set_transform()
        return self.gp

    def _set_transform(self):
        for g in self._all_gps():
            if self.normalize_output:
                g.set_output_transform(self.y_mean, self.y_std)
            else:
                g.set_output_transform(0.0, 1.0)

    # ---- BaseModel ----------------------------------------------------------------------------
    def _host_train(self, X, y, alloc=True):
        """the host half of train() (gaussian_process.py:89-104): normalisation and the constant mean;
        returns the device handle sized for the data (``alloc``), nothing uploaded yet"""
        if self.normalize_input:
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
        self.is_trained = False
        return self._ensure_gp(self.X.shape[0], self.X.shape[1]) if alloc else None

    def _host_train_raw(self, X, y, alloc=True):
        """_host_train on the caller's raw inputs (FabolasGP maps them through normalize() first)"""
        return self._host_train(X, y, alloc)

    def _adopt_fit(self, theta):
        """this model's device handle was fitted at theta by a batched pass (robo_gp_fit_batch)"""
        self.hypers = np.append(self.kernel.get_parameter_vector(), np.log(self.noise))
        self._set_transform()
        self._fitted_theta = np.array(theta, dtype=np.float64)
        self.is_trained = True

    @BaseModel._check_shapes_train
    def train(self, X, y, do_optimize=True):
        gp = self._host_train(X, y)
        self._upload()
        self._set_transform()

        if do_optimize:
            self.hypers = self.optimize()
            self.kernel.set_parameter_vector(self.hypers[:-1])
            self.noise = np.exp(self.hypers[-1])
        else:
            self.hypers = np.append(self.kernel.get_parameter_vector(), np.log(self.noise))
        logger.debug("GP Hyperparameters: " + str(self.hypers))

        try:
            theta = np.append(self.kernel.get_parameter_vector(), np.log(self.noise))
            self._fit_everywhere(theta)
        except np.linalg.LinAlgError:
            self.noise *= 10
            theta = np.append(self.kernel.get_parameter_vector(), np.log(self.noise))
            self._fit_everywhere(theta)
        self._fitted_theta = theta
        self.is_trained = True
        # what follows a train() in the reference's loop is an acquisition maximisation over a small candidate batch
        # (solver/bayesian_optimization.py:236-245, 500 candidates by default): the explicit inverse factor those batches
        # use is built on the device while the host prepares them.  Best effort: a failure here must not turn a
        # successful fit into an exception (the first small batch builds the inverse itself)
        for g in self._all_gps():
            try:
                g.prefetch_inverse()
            except Exception as e:        # noqa: BLE001
                logger.warning("prefetch of the explicit inverse factor failed (%s); it is built on first use", e)

    def get_noise(self):
        return self.noise

    def nll(self, theta):
        """Negative log marginal likelihood (+ prior) at theta; same failure protocol as the
        reference (1e25)."""
        theta = np.asarray(theta, dtype=np.float64)
        if np.any((-20 > theta) + (theta > 20)):
            return 1e25
        try:
            ll = self._device().fit(theta, self.mean)
        except np.linalg.LinAlgError:
            return 1e25
        if self.prior is not None:
            ll += self.prior.lnprob(theta)
        return -ll if np.isfinite(ll) else 1e25

    def grad_nll(self, theta):
        """Gradient of :meth:`nll` as the reference computes it (gaussian_process.py:168-191):
        ``-(0.5 einsum('ijk,ij', Kg, alpha alpha^T - K^-1) + prior.gradient(theta))`` -- on the
        device, without K^-1 or the (N, N, P) kernel-gradient tensor on the host.  Mirrored quirks:
        the noise entry is the derivative w.r.t. sigma^2 (the reference stacks an identity matrix as the
        noise "gradient", :178-182), and the prior term is whatever ``prior.gradient`` returns
        (zeros for DefaultPrior, default_priors.py:51-53)."""
        theta = np.asarray(theta, dtype=np.float64)
        _, g = self._device().grad_loglik(theta, self.mean)
        if self.prior is not None:
            g = g + self.prior.gradient(theta)
        return -g

    def _nll_with_gradient(self, theta):
        """(nll, d nll / d theta) with a CONSISTENT gradient for the optimiser: chain rule on the noise
        entry (d / d log sigma^2 = sigma^2 d / d sigma^2) and a central difference of the prior's
        lnprob (host scalar work) instead of ``prior.gradient`` -- one device pass for both."""
        theta = np.asarray(theta, dtype=np.float64)
        if np.any((-20 > theta) + (theta > 20)):
            return 1e25, np.zeros_like(theta)
        try:
            ll, g = self._device().grad_loglik(theta, self.mean)
        except np.linalg.LinAlgError:
            return 1e25, np.zeros_like(theta)
        g = g.copy()
        g[-1] *= np.exp(theta[-1])
        if self.prior is not None:
            ll += self.prior.lnprob(theta)
            h = 1e-6
            for p in range(theta.size):
                e = np.zeros_like(theta)
                e[p] = h
                with np.errstate(invalid="ignore"):     # -inf - -inf outside a tophat's support
                    d = (self.prior.lnprob(theta + e) - self.prior.lnprob(theta - e)) / (2 * h)
                if np.isfinite(d):
                    g[p] += d
        if not np.isfinite(ll) or not np.all(np.isfinite(g)):
            return 1e25, np.zeros_like(theta)
        return -ll, -g

    def optimize(self):
        p0 = np.append(self.kernel.get_parameter_vector(), np.log(self.noise))
        if self.use_gradients:
            # reference: optimize.minimize(self.nll, p0, method="BFGS", jac=self.grad_nll), whose
            # 3-tuple unpacking of the OptimizeResult cannot succeed (gaussian_process.py:207-210);
            # here the result's .x, with the consistent gradient above (DESIGN.md deviations)
            results = optimize.minimize(self._nll_with_gradient, p0, method="BFGS", jac=True)
            return results.x
        try:
            results = optimize.minimize(self.nll, p0, method='L-BFGS-B')
            theta = results.x
        except ValueError:
            logging.error("Could not find a valid hyperparameter configuration! Use initial configuration")
            theta = p0
        return theta

    def predict_variance(self, x1, X2):
        """Covariance between the test point x1 (1, D) and X2 (N, D) -> (N, 1)."""
        if not self.is_trained:
            raise Exception('Model has to be trained first!')
        x_ = np.concatenate((x1, X2))
        _, var = self.predict(x_, full_cov=True)
        return var[-1, :-1, np.newaxis]

    def _normalised(self, X_test):
        if self.normalize_input:
            return normalization.zero_one_normalization(X_test, self.lower, self.upper)[0]
        return X_test

    @BaseModel._check_shapes_predict
    def predict(self, X_test, full_cov=False, **kwargs):
        if not self.is_trained:
            raise Exception('Model has to be trained first!')
        self._materialise()
        Xn = self._normalised(X_test)
        if not full_cov:
            return self.gp.predict(Xn)          # transform + floor done on the device
        mu, cov = self.gp.predict_cov(Xn)
        # the reference clips the whole matrix, off-diagonals included (gaussian_process.py:290-294)
        eps = np.finfo(cov.dtype).eps
        cov = np.clip(cov, eps, np.inf)
        return mu, cov

    def predictive_gradients(self, X_test):
        """Gradients of the predictive mean and variance w.r.t. the inputs, in the convention the reference's
        callers index (GPy's): ``dmdx`` (M, D, 1), ``dvdx`` (M, D)  (robo/acquisition_functions/ei.py:80-85,
        lcb.py:66-68, robo/util/posterior_optimization.py:38-40,96-104).  No model of the reference implements
        this; here every point costs D + 1 right-hand sides of the device's blocked forward substitution
        (robo_gp_predict_grad)."""
        if not self.is_trained:
            raise Exception('Model has to be trained first!')
        self._materialise()
        X_test = np.asarray(X_test, dtype=np.float64)
        _, _, dm, dv = self.gp.predict_grad(self._normalised(X_test))
        if self.normalize_input:
            scale = 1.0 / (np.asarray(self.upper, dtype=np.float64) - np.asarray(self.lower, dtype=np.float64))
            dm, dv = dm * scale, dv * scale         # d x_normalised / d x
        return dm[:, :, np.newaxis], dv

    def sample_functions(self, X_test, n_funcs=1):
        """Draw n_funcs functions from the posterior at X_test -> (F, N)."""
        if not self.is_trained:
            raise Exception('Model has to be trained first!')
        self._materialise()
        mu, cov = self.gp.predict_cov(self._normalised(X_test))   # already un-normalised
        funcs = self.rng.multivariate_normal(mu, cov, size=n_funcs)
        return funcs[None, :] if funcs.ndim == 1 else funcs

    def get_incumbent(self):
        inc, inc_value = super(GaussianProcess, self).get_incumbent()
        if self.normalize_input:
            inc = normalization.zero_one_unnormalization(inc, self.lower, self.upper)
        if self.normalize_output:
            inc_value = normalization.zero_mean_unit_var_unnormalization(inc_value, self.y_mean, self.y_std)
        return inc, inc_value

    # ---- fused device paths used by robo_amd's acquisition functions ----------------------------
    def acquisition(self, kind, par, eta, X_test, want_values=True):
        """(values, max, argmax, flags) of a closed-form acquisition at raw candidates X_test."""
        if not self.is_trained:
            raise Exception('Model has to be trained first!')
        self._materialise()
        if isinstance(X_test, _lib.CandidateShards):
            vals, mx, am, _, flags = self._multi().acq(self._all_gps(), kind, par, eta, X_test, want_values)
            return vals, mx, am, flags
        if isinstance(X_test, _lib.Candidates):
            return self.gp.acq(kind, par, eta, X_test, want_values)
        Xn = self._normalised(X_test)
        if self.devices and np.asarray(Xn).shape[0] >= len(self.devices):
            # single-process multi-GPU: contiguous shards of the ONE candidate matrix, a replica of the fitted model on
            # every device, all devices at once; (max, index, flags) reduced with np.argmax's tie-break
            shards = self._shards_for(Xn)
            vals, mx, am, _, flags = self._multi().acq(self._all_gps(), kind, par, eta, shards, want_values)
            return vals, mx, am, flags
        return self.gp.acq(kind, par, eta, Xn, want_values)

"""Log-priors on log-scale GP hyper-parameters (host side, O(P) scalar work per call).

Behavioural mirrors of robo/priors/base_prior.py:75-395 and
robo/priors/default_priors.py:7-60, including the reference's quirks that change numbers
(SURVEY.md A.3 item 10): ``NormalPrior.lnprob`` returns a pdf, not a log-pdf
(base_prior.py:357); ``HorseshoePrior.lnprob`` is +inf at theta == 0 (:194-195);
``LognormalPrior.lnprob`` is ``lognorm.logpdf(theta, sigma, loc=mean)`` on the log-scale
theta (:278).  They are part of the posterior the reference's MCMC samples from, so they
are kept, not "fixed".
"""
import numpy as np
import scipy.stats as sps


def _rng(rng):
    return np.random.RandomState(np.random.randint(0, 10000)) if rng is None else rng


class BasePrior(object):
    def __init__(self, rng=None):
        self.rng = _rng(rng)

    def lnprob(self, theta):
        raise NotImplementedError

    def sample_from_prior(self, n_samples):
        raise NotImplementedError

    def gradient(self, theta):
        raise NotImplementedError


class TophatPrior(BasePrior):
    """0 inside [l_bound, u_bound] (log scale), -inf outside."""

    def __init__(self, l_bound, u_bound, rng=None):
        super(TophatPrior, self).__init__(rng)
        if not u_bound > l_bound:
            raise Exception("Upper bound of Tophat prior must be greater than the lower bound!")
        self.min, self.max = l_bound, u_bound

    def lnprob(self, theta):
        inside = not (np.any(theta < self.min) or np.any(theta > self.max))
        return 0 if inside else -np.inf

    def sample_from_prior(self, n_samples):
        return (self.min + self.rng.rand(n_samples) * (self.max - self.min))[:, np.newaxis]

    def gradient(self, theta):
        return np.zeros([theta.shape[0]])


class HorseshoePrior(BasePrior):
    """Spearmint's horseshoe approximation: log(log(1 + 3 (scale / exp(theta))^2))."""

    def __init__(self, scale=0.1, rng=None):
        super(HorseshoePrior, self).__init__(rng)
        self.scale = scale

    def lnprob(self, theta):
        if np.any(theta == 0.0):
            return np.inf
        return np.log(np.log(1 + 3.0 * (self.scale / np.exp(theta)) ** 2))

    def sample_from_prior(self, n_samples):
        lamda = np.abs(self.rng.standard_cauchy(size=n_samples))
        return np.log(np.abs(self.rng.randn() * lamda * self.scale))[:, np.newaxis]

    def gradient(self, theta):
        s2 = self.scale ** 2
        denom = (3 * s2 + np.exp(2 * theta)) * np.log(3 * s2 * np.exp(-2 * theta) + 1)
        return -(6 * s2) / denom


def _lognorm_logpdf(x, sigma, loc):
    """scipy.stats.lognorm.logpdf(x, sigma, loc=loc) in closed form (the generic scipy entry point costs
    ~60 us per call, which at small N is as much as the device's batched likelihood itself):
    -log(sigma y sqrt(2 pi)) - log(y)^2 / (2 sigma^2) with y = x - loc, -inf for y <= 0."""
    y = np.asarray(x, dtype=np.float64) - loc
    with np.errstate(divide="ignore", invalid="ignore"):
        ly = np.log(y)
        out = -(ly * ly) / (2.0 * sigma * sigma) - ly - np.log(sigma * np.sqrt(2.0 * np.pi))
    return np.where(y > 0, out, -np.inf)


class LognormalPrior(BasePrior):
    def __init__(self, sigma, mean=0, rng=None):
        super(LognormalPrior, self).__init__(rng)
        self.sigma, self.mean = sigma, mean

    def lnprob(self, theta):
        return sps.lognorm.logpdf(theta, self.sigma, loc=self.mean)   # sic: the mean is passed as loc

    def sample_from_prior(self, n_samples):
        return self.rng.lognormal(mean=self.mean, sigma=self.sigma, size=n_samples)[:, np.newaxis]

    def gradient(self, theta):
        return None


class NormalPrior(BasePrior):
    def __init__(self, sigma, mean=0, rng=None):
        super(NormalPrior, self).__init__(rng)
        self.sigma, self.mean = sigma, mean

    def lnprob(self, theta):
        return sps.norm.pdf(theta, scale=self.sigma, loc=self.mean)   # sic: pdf (base_prior.py:357)

    def sample_from_prior(self, n_samples):
        return self.rng.normal(loc=self.mean, scale=self.sigma, size=n_samples)[:, np.newaxis]

    def gradient(self, theta):
        s = self.sigma
        return (1 / (s * np.sqrt(2 * np.pi))) * (-theta / s ** 2 * np.exp(-theta ** 2 / (2 * s ** 2)))


class DefaultPrior(BasePrior):
    """theta = [log amp, log metric_1..D, log noise]: lognormal(0,1) on the amplitude, tophat
    [-10, 2] on the length scales, horseshoe(0.1) on the noise (default_priors.py:7-37)."""

    def __init__(self, n_dims, rng=None):
        super(DefaultPrior, self).__init__(rng)
        self.n_dims = n_dims
        self.tophat = TophatPrior(-10, 2, rng=self.rng)
        self.ln_prior = LognormalPrior(mean=0.0, sigma=1.0, rng=self.rng)
        self.horseshoe = HorseshoePrior(scale=0.1, rng=self.rng)

    def lnprob(self, theta):
        return (self.ln_prior.lnprob(theta[0]) + self.tophat.lnprob(theta[1:-1])
                + self.horseshoe.lnprob(theta[-1]))

    def lnprob_batch(self, thetas):
        """lnprob for a (k, P) batch of thetas in one vectorised evaluation (the ensemble sampler asks
        for half an ensemble at a time); the values of k calls of lnprob (to rounding: closed-form lognormal)."""
        thetas = np.atleast_2d(thetas)
        ls = thetas[:, 1:-1]
        top = np.where(np.any(ls < self.tophat.min, axis=1) | np.any(ls > self.tophat.max, axis=1), -np.inf, 0.0)
        noise = thetas[:, -1]
        with np.errstate(divide="ignore", over="ignore"):
            hs = np.log(np.log(1 + 3.0 * (self.horseshoe.scale / np.exp(noise)) ** 2))
        hs = np.where(noise == 0.0, np.inf, hs)
        return _lognorm_logpdf(thetas[:, 0], self.ln_prior.sigma, self.ln_prior.mean) + top + hs

    def sample_from_prior(self, n_samples):
        p0 = np.zeros([n_samples, self.n_dims])
        p0[:, 0] = self.ln_prior.sample_from_prior(n_samples)[:, 0]
        # one tophat draw per length-scale column, in column order (default_priors.py:47-49)
        for col in range(1, self.n_dims - 1):
            p0[:, col] = self.tophat.sample_from_prior(n_samples)[:, 0]
        p0[:, -1] = self.horseshoe.sample_from_prior(n_samples)[:, 0]
        return p0

    def gradient(self, theta):
        return np.zeros([theta.shape[0]])


class EnvPrior(BasePrior):
    """Prior of the Fabolas kernels (robo/priors/env_priors.py:8-80): theta = [log amp,
    log metric_1..n_ls, n_lr Bayesian-linear-regression parameters, log noise] with
    lognormal(mean=-2, sigma=1) on the amplitude, tophat [-10, 2] on the length scales,
    NormalPrior(sigma=1) on the regression parameters (whose ``lnprob`` is a pdf, see NormalPrior)
    and horseshoe(0.001) on the noise."""

    def __init__(self, n_dims, n_ls, n_lr, rng=None):
        super(EnvPrior, self).__init__(rng)
        self.n_dims, self.n_ls, self.n_lr = n_dims, n_ls, n_lr
        self.bayes_lin_prior = NormalPrior(sigma=1, mean=0, rng=self.rng)
        self.tophat = TophatPrior(-10, 2, rng=self.rng)
        self.ln_prior = LognormalPrior(mean=-2, sigma=1.0, rng=self.rng)
        self.horseshoe = HorseshoePrior(scale=0.001, rng=self.rng)

    def lnprob(self, theta):
        lp = self.ln_prior.lnprob(theta[0]) + self.tophat.lnprob(theta[1:self.n_ls + 1])
        for t in theta[self.n_ls + 1:self.n_ls + self.n_lr + 1]:
            lp += self.bayes_lin_prior.lnprob(t)
        return lp + self.horseshoe.lnprob(theta[-1])

    def sample_from_prior(self, n_samples):
        p0 = np.zeros([n_samples, self.n_dims])
        p0[:, 0] = self.ln_prior.sample_from_prior(n_samples)[:, 0]
        for col in range(1, self.n_ls + 1):
            p0[:, col] = self.tophat.sample_from_prior(n_samples)[:, 0]
        for col in range(self.n_ls + 1, self.n_ls + self.n_lr + 1):
            p0[:, col] = self.bayes_lin_prior.sample_from_prior(n_samples)[:, 0]
        p0[:, -1] = self.horseshoe.sample_from_prior(n_samples)[:, 0]
        return p0

    def gradient(self, theta):
        return np.zeros([theta.shape[0]])

"""Affine-invariant ensemble sampler (Goodman & Weare 2010 stretch move) with BATCHED
log-probability evaluation.

emcee (requirements.txt:3, ``>=2.1.0``, v2 API) is the sampler the reference drives in
robo/models/gaussian_process_mcmc.py:114-142:

    sampler = emcee.EnsembleSampler(n_walkers, ndim, lnprob)
    sampler.random_state = rng.get_state()
    pos, lnp, state = sampler.run_mcmc(p0, n_steps, rstate0=rng)
    sampler.chain[:, -1]

emcee is not installed here, and its one-walker-at-a-time ``lnprob(theta)`` callback is
exactly the pattern the device path wants to avoid: each half-ensemble update proposes
n_walkers/2 independent thetas, i.e. n_walkers/2 independent GP fits.  This class keeps the
call shape above (so GaussianProcessMCMC.train reads like the reference) but hands each
half-ensemble to ``lnprob_batch(thetas (k, ndim)) -> (k,)`` in one call, which
GaussianProcessMCMC maps onto robo_gp_loglik_batch.

Algorithm (emcee 2.x ``EnsembleSampler._propose_stretch``, restated from the paper): split
the walkers into two halves; for each half S (complement C), for every walker x_k in S draw
a partner c_j from C uniformly and z ~ g(z) ∝ 1/sqrt(z) on [1/a, a] (a = 2) via
z = ((a - 1) u + 1)^2 / a; propose q = c_j - z (c_j - x_k); accept with probability
min(1, z^(ndim-1) exp(lnp(q) - lnp(x_k))).  The draw order per half-step (rand for z, randint for the
partners, rand for the accept test) is emcee 2's, so with equal log-probabilities the chain equals the
one the reference produces (checked against a fixture made by the reference's GaussianProcessMCMC.train,
tests/ref_checks.py::check_ref_mcmc).
"""
import numpy as np


class EnsembleSampler(object):

    def __init__(self, nwalkers, dim, lnprob=None, lnprob_batch=None, a=2.0, device_chain=None):
        assert nwalkers % 2 == 0, "The number of walkers must be even."
        assert nwalkers >= 2 * dim, "The number of walkers needs to be at least twice the dimension"
        assert (lnprob is None) != (lnprob_batch is None)
        self.k, self.dim, self.a = int(nwalkers), int(dim), float(a)
        if lnprob_batch is None:
            def lnprob_batch(thetas, _f=lnprob):
                return np.array([_f(t) for t in thetas], dtype=np.float64)
        self._lnprob_batch = lnprob_batch
        # device_chain(p, lnp | None, N, u_stretch, partner, u_accept, a) -> (p, lnp, chain, lnps, accepted): the whole
        # run on the device (robo_gp_mcmc_run) with the random numbers drawn HERE, in the order below
        self._device_chain = device_chain
        self._random = np.random.RandomState()
        self.naccepted = np.zeros(self.k)
        self.iterations = 0
        self._chain = np.empty((self.k, 0, self.dim))
        self._lnprob = np.empty((self.k, 0))

    # emcee-style attributes ---------------------------------------------------------------
    @property
    def random_state(self):
        return self._random.get_state()

    @random_state.setter
    def random_state(self, state):
        try:
            self._random.set_state(state)
        except Exception:
            pass

    @property
    def chain(self):
        return self._chain

    @property
    def lnprobability(self):
        return self._lnprob

    @property
    def acceptance_fraction(self):
        return self.naccepted / max(self.iterations, 1)

    def _eval(self, thetas):
        lp = np.asarray(self._lnprob_batch(thetas), dtype=np.float64)
        if np.any(np.isnan(lp)):
            raise ValueError("lnprob returned NaN.")
        return lp

    def run_mcmc(self, pos0, N, rstate0=None, lnprob0=None):
        """-> (pos (k, dim), lnprob (k,), random state)"""
        if rstate0 is not None:
            # emcee 2 hands rstate0 to RandomState.set_state inside a try/except: a state TUPLE is applied, the
            # RandomState OBJECT the reference passes (gaussian_process_mcmc.py:126-135) is silently ignored, so
            # the stream set through ``sampler.random_state = rng.get_state()`` (:117) simply continues from
            # the burn-in run into the chain run.  Same here.
            self.random_state = rstate0
        p = np.array(pos0, dtype=np.float64)
        assert p.shape == (self.k, self.dim)
        draws = None
        if self._device_chain is not None:
            done, draws = self._run_on_device(p, N, lnprob0)
            if done is not None:
                return done
            # (the device declined -- e.g. half an ensemble does not fit its batch workspace: the host loop below
            # consumes the random numbers that were already drawn, in the same order)
        lnp = self._eval(p) if lnprob0 is None else np.array(lnprob0, dtype=np.float64)
        if np.any(np.isinf(lnp) & (lnp > 0)):
            raise ValueError("The initial lnprob was +inf.")
        chain = np.empty((self.k, N, self.dim))
        lnps = np.empty((self.k, N))
        half = self.k // 2
        first, second = slice(half), slice(half, self.k)
        for it in range(N):
            for S0, S1 in ((first, second), (second, first)):
                s, c = p[S0], p[S1]
                ns, nc = s.shape[0], c.shape[0]
                hh = 0 if S0 is first else 1
                u_z = self._random.rand(ns) if draws is None else draws[0][it, hh]
                zz = ((self.a - 1.0) * u_z + 1.0) ** 2.0 / self.a
                rint = self._random.randint(nc, size=(ns,)) if draws is None else draws[1][it, hh]
                q = c[rint] - zz[:, None] * (c[rint] - s)
                newlnp = self._eval(q)
                lnpdiff = (self.dim - 1.0) * np.log(zz) + newlnp - lnp[S0]
                accept = lnpdiff > np.log(self._random.rand(ns) if draws is None else draws[2][it, hh])
                if np.any(accept):
                    idx = np.arange(self.k)[S0][accept]
                    p[idx] = q[accept]
                    lnp[idx] = newlnp[accept]
                    self.naccepted[idx] += 1
            chain[:, it] = p
            lnps[:, it] = lnp
            self.iterations += 1
        self._chain = np.concatenate((self._chain, chain), axis=1)
        self._lnprob = np.concatenate((self._lnprob, lnps), axis=1)
        return p, lnp, self.random_state

    def _run_on_device(self, p, N, lnprob0):
        """The same chain with every half-step on the device.  A stretch move's random numbers do not depend on the
        state of the chain, so they are drawn up front -- per half-step rand (z), randint (partners), rand (accept
        test), the order of the loop in run_mcmc -- and the stream ends where it would have ended."""
        half = self.k // 2
        try:
            # one library call instead of 6 N NumPy calls: same numbers, same final state of the stream (robo_mcmc_draws)
            from robo_amd import _lib
            uz, pa, ua = _lib.mcmc_draws(self._random, N, half)
        except Exception:
            uz = np.empty((N, 2, half))
            ua = np.empty((N, 2, half))
            pa = np.empty((N, 2, half), dtype=np.int32)
            for it in range(N):
                for h in range(2):
                    uz[it, h] = self._random.rand(half)
                    pa[it, h] = self._random.randint(half, size=(half,))
                    ua[it, h] = self._random.rand(half)
        if lnprob0 is not None and np.any(np.isinf(lnprob0) & (np.asarray(lnprob0) > 0)):
            raise ValueError("The initial lnprob was +inf.")
        res = self._device_chain(p, lnprob0, N, uz, pa, ua, self.a)
        if res is None:
            return None, (uz, pa, ua)
        p, lnp, chain, lnps, acc = res
        self.naccepted += acc
        self.iterations += N
        self._chain = np.concatenate((self._chain, chain), axis=1)
        self._lnprob = np.concatenate((self._lnprob, lnps), axis=1)
        return (p, lnp, self.random_state), None

    def reset(self):
        self.naccepted[:] = 0
        self.iterations = 0
        self._chain = np.empty((self.k, 0, self.dim))
        self._lnprob = np.empty((self.k, 0))

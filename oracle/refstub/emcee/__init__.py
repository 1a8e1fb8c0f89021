"""Test-only stand-in for ``emcee`` 2.x (TEST INFRASTRUCTURE -- never imported by robo_amd).

``emcee>=2.1.0`` (requirements.txt:3) is not installed here.  The reference drives the v2 API:

    sampler = emcee.EnsembleSampler(nwalkers, dim, lnprob)           gaussian_process_mcmc.py:114
    sampler.random_state = rng.get_state()                          :117
    pos, lnprob, state = sampler.run_mcmc(p0, n, rstate0=rng)        :126-135
    sampler.chain[:, -1]                                             :142
    zb, lmb, _ = sampler.run_mcmc(restarts, 50)                      information_gain.py:139-142

This module restates emcee 2.2's ``EnsembleSampler`` (Goodman & Weare stretch move, a = 2) with
its draw order: per iteration, for each half S of the ensemble (complement C):
``zz = ((a-1) rand(|S|) + 1)^2 / a``; ``rint = randint(|C|, size=|S|)``;
``q = C[rint] - zz (C[rint] - S)``; one lnprob call per proposed walker, in order;
``accept = (dim-1) log zz + lnp(q) - lnp(s) > log(rand(|S|))``.  The ``random_state`` setter
swallows anything ``RandomState.set_state`` rejects -- so a RandomState OBJECT passed as
``rstate0`` (what the reference does) is silently ignored, as in emcee 2.
"""
import numpy as np

__version__ = "2.2.1-refstub"


class EnsembleSampler(object):

    def __init__(self, nwalkers, dim, lnpostfn, a=2.0, args=(), kwargs=None, **unused):
        assert nwalkers % 2 == 0, "The number of walkers must be even."
        assert nwalkers >= 2 * dim, \
            "The number of walkers needs to be more than twice the dimension of your parameter space."
        self.k, self.dim, self.a = int(nwalkers), int(dim), float(a)
        self.lnprobfn = lnpostfn
        self.args, self.kwargs = tuple(args), dict(kwargs or {})
        self._random = np.random.mtrand.RandomState()
        self.reset()

    def reset(self):
        self.naccepted = np.zeros(self.k)
        self.iterations = 0
        self._chain = np.empty((self.k, 0, self.dim))
        self._lnprob = np.empty((self.k, 0))

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
    def flatchain(self):
        s = self._chain.shape
        return self._chain.reshape(s[0] * s[1], s[2])

    @property
    def lnprobability(self):
        return self._lnprob

    @property
    def acceptance_fraction(self):
        return self.naccepted / self.iterations

    def _get_lnprob(self, pos):
        p = np.asarray(pos)
        if np.any(np.isinf(p)):
            raise ValueError("At least one parameter value was infinite.")
        if np.any(np.isnan(p)):
            raise ValueError("At least one parameter value was NaN.")
        lnprob = np.array([float(self.lnprobfn(p[i], *self.args, **self.kwargs)) for i in range(len(p))])
        if np.any(np.isnan(lnprob)):
            raise ValueError("lnprob returned NaN.")
        return lnprob

    def _propose_stretch(self, p0, p1, lnprob0):
        s = np.atleast_2d(p0)
        Ns = len(s)
        c = np.atleast_2d(p1)
        Nc = len(c)
        zz = ((self.a - 1.) * self._random.rand(Ns) + 1) ** 2. / self.a
        rint = self._random.randint(Nc, size=(Ns,))
        q = c[rint] - zz[:, np.newaxis] * (c[rint] - s)
        newlnprob = self._get_lnprob(q)
        lnpdiff = (self.dim - 1.) * np.log(zz) + newlnprob - lnprob0
        accept = (lnpdiff > np.log(self._random.rand(len(lnpdiff))))
        return q, newlnprob, accept

    def sample(self, p0, lnprob0=None, rstate0=None, iterations=1):
        self.random_state = rstate0
        p = np.array(p0)
        halfk = int(self.k / 2)
        lnprob = lnprob0
        if lnprob is None:
            lnprob = self._get_lnprob(p)
        if np.any(np.isnan(lnprob)):
            raise ValueError("The initial lnprob was NaN.")
        N = int(iterations)
        self._chain = np.concatenate((self._chain, np.zeros((self.k, N, self.dim))), axis=1)
        self._lnprob = np.concatenate((self._lnprob, np.zeros((self.k, N))), axis=1)
        i0 = self._chain.shape[1] - N
        for i in range(N):
            self.iterations += 1
            first, second = slice(halfk), slice(halfk, self.k)
            for S0, S1 in [(first, second), (second, first)]:
                q, newlnp, acc = self._propose_stretch(p[S0], p[S1], lnprob[S0])
                if np.any(acc):
                    lnprob[S0][acc] = newlnp[acc]
                    p[S0][acc] = q[acc]
                    self.naccepted[S0][acc] += 1
            self._chain[:, i0 + i, :] = p
            self._lnprob[:, i0 + i] = lnprob
            yield p, lnprob, self.random_state

    def run_mcmc(self, pos0, N, rstate0=None, lnprob0=None, **kwargs):
        results = None
        for results in self.sample(pos0, lnprob0, rstate0, iterations=N, **kwargs):
            pass
        return results

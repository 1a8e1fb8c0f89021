"""robo/initial_design/init_latin_hypercube_sampling.py:4-37 -- same signature and RNG call sequence."""
import numpy as np


def _rng(rng):
    return np.random.RandomState(np.random.randint(0, 10000)) if rng is None else rng


def init_latin_hypercube_sampling(lower, upper, n_points, rng=None):
    """(n_points, D) Latin hypercube: one uniform draw per stratum, then an independent
    shuffle per dimension (same RNG call sequence as the reference, :27-36)."""
    rng = _rng(rng)
    d = lower.shape[0]
    edges = np.array([np.linspace(lower[i], upper[i], n_points + 1) for i in range(d)])
    lo, hi = edges[:, :-1], edges[:, 1:]
    pts = lo + rng.uniform(0, 1, lo.shape) * (hi - lo)
    for i in range(d):
        rng.shuffle(pts[i, :])
    return pts.T

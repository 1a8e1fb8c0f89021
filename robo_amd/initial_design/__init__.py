"""Initial designs (host side, O(N D)): same signatures and draw order as
robo/initial_design/init_random_uniform.py:4-30 and
init_latin_hypercube_sampling.py:4-37, so seeded runs pick the same points."""
import numpy as np


def _rng(rng):
    return np.random.RandomState(np.random.randint(0, 10000)) if rng is None else rng


def init_random_uniform(lower, upper, n_points, rng=None):
    """(n_points, D) uniform in the box, drawn ROW BY ROW like the reference (:29-30)."""
    rng = _rng(rng)
    d = lower.shape[0]
    out = np.empty((n_points, d))
    for i in range(n_points):
        out[i] = rng.uniform(lower, upper, d)
    return out


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

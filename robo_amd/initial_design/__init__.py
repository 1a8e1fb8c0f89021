"""Initial designs (host side, O(N D)): same signatures and draw order as
robo/initial_design/init_random_uniform.py:4-30 and
init_latin_hypercube_sampling.py:4-37, so seeded runs pick the same points."""
import numpy as np


def _rng(rng):
    return np.random.RandomState(np.random.randint(0, 10000)) if rng is None else rng


def init_random_uniform(lower, upper, n_points, rng=None):
    """(n_points, D) uniform in the box: the numbers the reference draws row by row (:29-30) -- a legacy RandomState
    fills an array in C order from one sequential stream, so ONE (n_points, D) call yields the same values, bit for bit,
    as n_points calls of size D (pinned by tests/test_host_logic.py against the reference's own loop); the Python loop
    was 2.3 ms of the 500-candidate maximisation of a BO iteration."""
    rng = _rng(rng)
    return rng.uniform(lower, upper, (int(n_points), lower.shape[0]))


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


def init_grid(lower, upper, n_points):
    """(n_points ** D, D) full grid with n_points levels per dimension, end points included, in the row order of
    np.meshgrid's default 'xy' indexing (robo/initial_design/init_grid.py:23-30)."""
    levels = [np.linspace(lo, hi, n_points) for lo, hi in zip(lower, upper)]
    return np.stack([axis.ravel() for axis in np.meshgrid(*levels)], axis=1).astype(np.float64)


def init_random_normal(lower, upper, n_points, mean=None, std=None, rng=None):
    """(n_points, D) points from N(mean_d, std_d) per dimension, clipped to the box; defaults: the centre of the box and
    std 0.1; one draw of n_points numbers per dimension, dimension by dimension (init_random_normal.py:30-43)."""
    rng = _rng(rng)
    d = lower.shape[0]
    mean = 0.5 * (upper + lower) if mean is None else mean
    std = np.full(d, 0.1) if std is None else std
    cols = [np.clip(rng.normal(mean[i], std[i], n_points), lower[i], upper[i]) for i in range(d)]
    return np.stack(cols, axis=1)

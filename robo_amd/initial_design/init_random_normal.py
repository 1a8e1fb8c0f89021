"""robo/initial_design/init_random_normal.py:4-43 -- same signature, defaults and draw order."""
import numpy as np


def _rng(rng):
    return np.random.RandomState(np.random.randint(0, 10000)) if rng is None else rng


def init_random_normal(lower, upper, n_points, mean=None, std=None, rng=None):
    """(n_points, D) points from N(mean_d, std_d) per dimension, clipped to the box; defaults: the centre of the box and
    std 0.1; one draw of n_points numbers per dimension, dimension by dimension (init_random_normal.py:30-43)."""
    rng = _rng(rng)
    d = lower.shape[0]
    mean = 0.5 * (upper + lower) if mean is None else mean
    std = np.full(d, 0.1) if std is None else std
    cols = [np.clip(rng.normal(mean[i], std[i], n_points), lower[i], upper[i]) for i in range(d)]
    return np.stack(cols, axis=1)

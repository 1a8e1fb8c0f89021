"""robo/initial_design/init_random_uniform.py:4-30 -- same signature and draw order, one block call."""
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

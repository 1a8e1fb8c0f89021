"""robo/initial_design/init_grid.py:4-30 -- same signature and row order."""
import numpy as np


def init_grid(lower, upper, n_points):
    """(n_points ** D, D) full grid with n_points levels per dimension, end points included, in the row order of
    np.meshgrid's default 'xy' indexing (robo/initial_design/init_grid.py:23-30)."""
    levels = [np.linspace(lo, hi, n_points) for lo, hi in zip(lower, upper)]
    return np.stack([axis.ravel() for axis in np.meshgrid(*levels)], axis=1).astype(np.float64)

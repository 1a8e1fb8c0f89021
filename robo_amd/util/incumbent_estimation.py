"""Incumbent = training configuration with the lowest PREDICTED mean after projecting the
environmental column to ``proj_value`` (robo/util/incumbent_estimation.py:4-14)."""
import numpy as np


def projected_incumbent_estimation(model, X, proj_value=1):
    X_projected = np.concatenate((X, np.full((X.shape[0], 1), float(proj_value))), axis=1)
    m, _ = model.predict(X_projected)
    best = np.argmin(m)
    return X_projected[best], m[best]

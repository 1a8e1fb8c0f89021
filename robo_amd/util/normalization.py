"""Affine input/output maps used around the GP (host-side prologue/epilogue).

Same call signatures and return tuples as the four helpers in
robo/util/normalization.py:4-32, so reference code calling them keeps working:
``zero_one_normalization`` (box -> [0,1]^D), ``zero_one_unnormalization``,
``zero_mean_unit_var_normalization`` (z-score), ``zero_mean_unit_var_unnormalization``.
"""
import numpy as np


class Affine(object):
    """x -> (x - shift) / scale and back; shift/scale broadcast over rows."""

    __slots__ = ("shift", "scale")

    def __init__(self, shift, scale):
        self.shift, self.scale = shift, scale

    def forward(self, x):
        return np.true_divide(x - self.shift, self.scale)

    def backward(self, u):
        return self.shift + self.scale * u


def zero_one_normalization(X, lower=None, upper=None):
    lower = X.min(axis=0) if lower is None else lower
    upper = X.max(axis=0) if upper is None else upper
    return Affine(lower, upper - lower).forward(X), lower, upper


def zero_one_unnormalization(X_normalized, lower, upper):
    return Affine(lower, upper - lower).backward(X_normalized)


def zero_mean_unit_var_normalization(X, mean=None, std=None):
    mean = X.mean(axis=0) if mean is None else mean
    std = X.std(axis=0) if std is None else std
    return Affine(mean, std).forward(X), mean, std


def zero_mean_unit_var_unnormalization(X_normalized, mean, std):
    # the reference evaluates X * std + mean (robo/util/normalization.py:32); keep that order
    return X_normalized * std + mean

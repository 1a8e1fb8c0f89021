"""Model plugin surface -- the contract of robo/models/base_model.py:5-106.

``train(X, y)``, ``update(X, y)``, ``predict(X_test) -> (mean (M,), var (M,))``,
``get_incumbent() -> (x, y)``, ``get_json_data()``.  Any object with these methods (the
reference's own BaseModel subclasses included) is accepted by robo_amd's acquisition
functions, maximisers and solver.
"""
import abc
import functools

import numpy as np


class BaseModel(object):
    __metaclass__ = abc.ABCMeta

    def __init__(self):
        self.X = None
        self.y = None

    @abc.abstractmethod
    def train(self, X, y):
        """Fit on X (N, D), y (N,)."""

    def update(self, X, y):
        """Append data and retrain (robo/models/base_model.py:30-45)."""
        self.train(np.append(self.X, X, axis=0), np.append(self.y, y, axis=0))

    @abc.abstractmethod
    def predict(self, X_test):
        """-> (mean (M,), var (M,)) at X_test (M, D)."""

    # shape guards: the reference uses bare asserts (base_model.py:66-79); so do we
    def _check_shapes_train(func):
        @functools.wraps(func)
        def guarded(self, X, y, *args, **kwargs):
            assert X.shape[0] == y.shape[0]
            assert len(X.shape) == 2
            assert len(y.shape) == 1
            return func(self, X, y, *args, **kwargs)
        return guarded

    def _check_shapes_predict(func):
        @functools.wraps(func)
        def guarded(self, X, *args, **kwargs):
            assert len(X.shape) == 2
            return func(self, X, *args, **kwargs)
        return guarded

    def get_json_data(self):
        return {'X': None if self.X is None else self.X.tolist(),
                'y': None if self.y is None else self.y.tolist(),
                'hyperparameters': ""}

    def get_incumbent(self):
        """Best observed point and its value (argmin of y), base_model.py:94-106."""
        best = np.argmin(self.y)
        return self.X[best], self.y[best]

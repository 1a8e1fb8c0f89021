from robo_amd.fmin.bayesian_optimization import bayesian_optimization  # noqa: F401
from robo_amd.fmin.entropy_search import entropy_search  # noqa: F401
from robo_amd.fmin.fabolas import fabolas  # noqa: F401
from robo_amd.fmin.random_search import random_search  # noqa: F401

"""robo/maximizers/base_maximizer.py under its own module path."""
from robo_amd.maximizers.random_sampling import BaseMaximizer  # noqa: F401

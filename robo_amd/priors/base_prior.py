"""robo/priors/base_prior.py under its own module path: the prior building blocks (robo_amd/priors/priors.py)."""
from robo_amd.priors.priors import BasePrior, HorseshoePrior, LognormalPrior, NormalPrior, TophatPrior  # noqa: F401

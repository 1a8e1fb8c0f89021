"""robo/priors/env_priors.py under its own module path."""
from robo_amd.priors.priors import (BasePrior, EnvPrior, HorseshoePrior, LognormalPrior, NormalPrior,  # noqa: F401
                                    TophatPrior)

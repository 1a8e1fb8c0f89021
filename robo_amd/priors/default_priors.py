"""robo/priors/default_priors.py under its own module path (it imports the building blocks too: callers take
``TophatPrior`` from here, test/test_models/test_gaussian_process.py:8)."""
from robo_amd.priors.priors import (BasePrior, DefaultPrior, HorseshoePrior, LognormalPrior,  # noqa: F401
                                    NormalPrior, TophatPrior)

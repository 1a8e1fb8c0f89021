from robo_amd.priors.priors import (BasePrior, TophatPrior, HorseshoePrior, LognormalPrior,  # noqa: F401
                                    NormalPrior, DefaultPrior, EnvPrior)

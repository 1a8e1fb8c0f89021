from robo_amd.models.base_model import BaseModel  # noqa: F401
from robo_amd.models.gaussian_process import GaussianProcess  # noqa: F401
from robo_amd.models.gaussian_process_mcmc import GaussianProcessMCMC  # noqa: F401
from robo_amd.models.fabolas_gp import FabolasGP, FabolasGPMCMC  # noqa: F401

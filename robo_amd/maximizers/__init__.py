from robo_amd.maximizers.random_sampling import (BaseMaximizer, DeviceRandomSampling, DeviceSobolSampling,  # noqa: F401
                                                    RandomSampling)
from robo_amd.maximizers.scipy_optimizer import SciPyOptimizer  # noqa: F401
from robo_amd.maximizers.differential_evolution import DifferentialEvolution  # noqa: F401
from robo_amd.maximizers.grid_search import GridSearch  # noqa: F401

from robo_amd.maximizers.random_sampling import (BaseMaximizer, DeviceRandomSampling, DeviceSobolSampling,  # noqa: F401
                                                    RandomSampling)

from robo_amd.maximizers.random_sampling import BaseMaximizer, DeviceRandomSampling, RandomSampling  # noqa: F401

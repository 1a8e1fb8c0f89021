from robo_amd.maximizers.random_sampling import BaseMaximizer, RandomSampling  # noqa: F401

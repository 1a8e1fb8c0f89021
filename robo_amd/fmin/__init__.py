from robo_amd.fmin.bayesian_optimization import bayesian_optimization  # noqa: F401

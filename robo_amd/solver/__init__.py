from robo_amd.solver.bayesian_optimization import BayesianOptimization  # noqa: F401

from robo_amd.solver.bayesian_optimization import BaseSolver, BayesianOptimization  # noqa: F401

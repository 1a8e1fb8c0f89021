"""robo/solver/base_solver.py under its own module path."""
from robo_amd.solver.bayesian_optimization import BaseSolver  # noqa: F401

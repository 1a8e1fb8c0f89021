"""One-dimensional grid maximiser (robo/maximizers/grid_search.py:7-48): ``resolution`` equally spaced points between the
bounds, end points included, the one with the highest acquisition value.  The reference scores the grid point by point in
a Python loop; here the whole grid is ONE batched call of the acquisition function (one device call for robo_amd's
closed-form functions and, where the function offers ``argmax``, only the winning index comes back)."""
import numpy as np

from robo_amd.maximizers.random_sampling import BaseMaximizer


class GridSearch(BaseMaximizer):

    def __init__(self, objective_function, lower, upper, resolution=1000):
        self.resolution = resolution
        if lower.shape[0] > 1:
            raise RuntimeError("Grid search works just for one dimensional functions")
        super(GridSearch, self).__init__(objective_function, lower, upper)

    def maximize(self):
        grid = np.linspace(self.lower[0], self.upper[0], self.resolution).reshape(self.resolution, 1)
        if hasattr(self.objective_func, "argmax"):
            return grid[self.objective_func.argmax(grid)]
        values = np.asarray(self.objective_func(grid), dtype=np.float64).reshape(-1)
        return grid[int(values.argmax())]

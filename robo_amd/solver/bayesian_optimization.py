"""Sequential Bayesian-optimisation loop (host control plane).

robo_amd's models and acquisition functions drop into the reference's own
``robo.solver.bayesian_optimization.BayesianOptimization`` unchanged (tests/test_dropin.py
drives exactly that).  This module is the george-free equivalent for installations where
the ``robo`` package cannot be imported; it keeps the constructor, ``run`` / ``choose_next``
and the bookkeeping lists of robo/solver/bayesian_optimization.py:16-261
(``time_overhead``, ``time_func_evals``, ``incumbents``, ``incumbents_values``,
``runtime``, per-iteration ``robo_iter_%d.json``).
"""
import json
import logging
import os
import time

import numpy as np

from robo_amd.initial_design import init_random_uniform

logger = logging.getLogger(__name__)


class BaseSolver(object):
    """What robo/solver/base_solver.py:12-139 gives every solver: the four collaborators as attributes, a run directory
    with results.csv / results.json, and the per-iteration JSON record put together from solver, model, task and
    acquisition function.  Host control plane; kept because callers subclass it and call its helpers."""

    def __init__(self, acquisition_func=None, model=None, maximize_func=None, task=None, save_dir=None):
        self.acquisition_func, self.model, self.maximize_func = acquisition_func, model, maximize_func
        self.task, self.save_dir = task, save_dir
        if save_dir is not None:
            self.create_save_dir()

    def create_save_dir(self):
        os.makedirs(self.save_dir, exist_ok=True)                # an existing directory is reused
        self.output_file, self.output_file_json = (open(os.path.join(self.save_dir, "results." + ext), "w")
                                                   for ext in ("csv", "json"))
        self.csv_writer = self.json_writer = None

    def get_observations(self):
        # the reference returns ``self.X, self.Y`` (:62-63) although BayesianOptimization keeps its targets in ``y``:
        # whichever of the two this solver holds
        return self.X, getattr(self, "Y", getattr(self, "y", None))

    def get_model(self):
        if self.model is None:
            logger.info("No model trained yet!")
        return self.model

    def run(self, num_iterations=10, X=None, y=None):
        """the optimisation loop -> (incumbent, incumbent value)"""

    def choose_next(self, X=None, y=None):
        """-> the next point to evaluate"""

    def get_json_data(self, it):
        """the solver's share of an iteration record; expects ``time_overhead``, ``time_func_eval``, ``incumbent``,
        ``incumbent_value`` and ``time_start`` on the solver (:112-123)"""
        record = dict(iteration=it, runtime=time.time() - self.time_start)
        record["optimization_overhead"], record["time_func_eval"] = self.time_overhead[it], self.time_func_eval[it]
        record["incumbent"], record["incumbent_fval"] = self.incumbent.tolist(), self.incumbent_value.tolist()
        return record

    def save_json(self, it, **kwargs):
        """one line of results.json per call (key spelling 'Acquisiton' as in :131-135)"""
        parts = (("Solver", self.get_json_data(it)), ("Model", self.model.get_json_data()),
                 ("Task", self.task.get_json_data()), ("Acquisiton", self.acquisition_func.get_json_data()))
        self.output_file_json.write(json.dumps(dict(parts)) + "\n")


class BayesianOptimization(BaseSolver):

    def __init__(self, objective_func, lower, upper, acquisition_func, model, maximize_func,
                 initial_design=init_random_uniform, initial_points=3, output_path=None, train_interval=1,
                 n_restarts=1, rng=None):
        self.rng = np.random.RandomState(np.random.randint(100000)) if rng is None else rng
        self.model = model
        self.acquisition_func = acquisition_func
        self.maximize_func = maximize_func
        self.start_time = time.time()
        self.initial_design = initial_design
        self.objective_func = objective_func
        self.X = None
        self.y = None
        self.time_func_evals = []
        self.time_overhead = []
        self.train_interval = train_interval
        self.lower = lower
        self.upper = upper
        self.output_path = output_path
        self.time_start = None
        self.incumbents = []
        self.incumbents_values = []
        self.n_restarts = n_restarts
        self.init_points = initial_points
        self.runtime = []

    def _spmd(self):
        """(rank, world) when this loop runs as one process per GPU with a shard switched on in one of its objects
        (explicit opt-in flags only: nothing here touches a communicator otherwise), else None"""
        acq = self.acquisition_func
        if not (getattr(self.maximize_func, "shard", False) or getattr(self.model, "sample_shard", False) or
                getattr(acq, "sample_shard", False) or getattr(acq, "shard", False)):
            return None
        from robo_amd import sharding
        _, rank, world = sharding.dist_info()
        return (rank, world) if world > 1 else None

    def _evaluate(self, x):
        """ONE objective evaluation per iteration (robo/solver/bayesian_optimization.py:123,174), also when every rank of
        a one-process-per-GPU job runs this loop in lock step: rank 0 evaluates, the value travels to the others (an
        expensive or non-deterministic objective must not run world-size times, nor give the ranks different data)."""
        spmd = self._spmd()
        if spmd is None:
            return self.objective_func(x)
        from robo_amd import sharding
        rank, _ = spmd
        value, failed, err = 0.0, 0.0, None
        if rank == 0:
            try:
                value = float(self.objective_func(x))
            except Exception as e:      # noqa: BLE001 -- the other ranks must leave the exchange too
                failed, err = 1.0, e
        row = sharding.allgather_rows([value, failed])[0]
        if row[1] != 0.0:
            if err is not None:
                raise err
            raise RuntimeError("the objective function failed on rank 0")
        return float(row[0])

    def _replicate_representer_points(self):
        """One process per GPU with a shard switched on somewhere in this loop: an entropy-search acquisition (or the
        per-sample estimators of a marginalised one) draws its representer points from an OS-seeded sampler, as in the
        reference (information_gain.py:139-142) -- they differ from rank to rank unless ``shard`` is set on it too, and ranks
        that score candidates against different p_min beliefs pick different points.  Forgetting that flag must not split
        the job: it is switched on here (rank 0's points on every rank), with a warning."""
        acq = self.acquisition_func
        for a in [acq] + list(getattr(acq, "estimators", [])):
            if hasattr(a, "sample_representer_points") and not getattr(a, "shard", False):
                logger.warning("%s.shard was off in a one-process-per-GPU run: switched on (rank 0's representer points "
                               "on every rank)", type(a).__name__)
                a.shard = True

    def _record_incumbent(self, X, y):
        best = int(np.argmin(y))
        self.incumbents.append(np.asarray(X[best]).tolist())
        self.incumbents_values.append(y[best])
        self.runtime.append(time.time() - self.start_time)

    def run(self, num_iterations=10, X=None, y=None):
        """-> (incumbent, incumbent value) after num_iterations evaluations in total."""
        self.time_start = time.time()
        if X is None and y is None:
            Xl, yl = [], []
            t0 = time.time()
            init = self.initial_design(self.lower, self.upper, self.init_points, rng=self.rng)
            design_overhead = (time.time() - t0) / self.init_points
            for i, x in enumerate(init):
                logger.info("Evaluate: %s", x)
                t0 = time.time()
                new_y = self._evaluate(x)
                Xl.append(x)
                yl.append(new_y)
                self.time_func_evals.append(time.time() - t0)
                self.time_overhead.append(design_overhead)
                self._record_incumbent(Xl, yl)
                if self.output_path is not None:
                    self.save_output(i)
            self.X, self.y = np.array(Xl), np.array(yl)
        else:
            self.X, self.y = X, y

        for it in range(self.init_points, num_iterations):
            logger.info("Start iteration %d ... ", it)
            t0 = time.time()
            new_x = self.choose_next(self.X, self.y, do_optimize=(it % self.train_interval == 0))
            self.time_overhead.append(time.time() - t0)
            logger.info("Optimization overhead was %f seconds", self.time_overhead[-1])
            t0 = time.time()
            new_y = self._evaluate(new_x)
            self.time_func_evals.append(time.time() - t0)
            logger.info("Configuration %s achieved a performance of %f", str(new_x), new_y)
            self.X = np.append(self.X, new_x[None, :], axis=0)
            self.y = np.append(self.y, new_y)
            self._record_incumbent(self.X, self.y)
            if self.output_path is not None:
                self.save_output(it)
        return self.incumbents[-1], self.incumbents_values[-1]

    def choose_next(self, X=None, y=None, do_optimize=True):
        """Train the model, update the acquisition function, maximise it -> next point (D,)."""
        if (X is None and y is None) or X.shape[0] == 1:
            # a GP needs at least two points (bayesian_optimization.py:226-231)
            return self.initial_design(self.lower, self.upper, 1, rng=self.rng)[0, :]
        try:
            t0 = time.time()
            self.model.train(X, y, do_optimize=do_optimize)
            logger.info("Time to train the model: %f", (time.time() - t0))
        except Exception:
            logger.error("Model could not be trained!")
            raise
        if self._spmd() is not None:
            self._replicate_representer_points()
        self.acquisition_func.update(self.model)
        t0 = time.time()
        x = self.maximize_func.maximize()
        logger.info("Time to maximize the acquisition function: %f", (time.time() - t0))
        return x

    def save_output(self, it):
        data = {"optimization_overhead": self.time_overhead[it], "runtime": self.runtime[it],
                "incumbent": self.incumbents[it], "incumbents_value": self.incumbents_values[it],
                "time_func_eval": self.time_func_evals[it], "iteration": it}
        with open(os.path.join(self.output_path, "robo_iter_%d.json" % it), "w") as fh:
            json.dump(data, fh)

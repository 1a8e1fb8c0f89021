"""The public surface on the path: every class / function of the reference that SURVEY.md section 8 puts on (or next to)
the hot path, with its parameter ORDER, parameter NAMES, DEFAULTS and public METHODS, as recorded from the reference itself
(tests/golden/ref_api_surface.json, written by tests/golden/make_golden_ref.py api_surface) -- robo_amd's object of the same
name must accept the same call: the reference's parameters lead, in order, with the same defaults; extra parameters only
behind them.  A caller who switches the import sees no TypeError and no changed default."""
import importlib
import inspect
import json
import os

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))

# where the reference's object lives in robo_amd (module path; the name is the same)
HOME = {
    "robo.models.gaussian_process": "robo_amd.models", "robo.models.gaussian_process_mcmc": "robo_amd.models",
    "robo.models.fabolas_gp": "robo_amd.models", "robo.models.base_model": "robo_amd.models.base_model",
    "robo.acquisition_functions.ei": "robo_amd.acquisition_functions",
    "robo.acquisition_functions.log_ei": "robo_amd.acquisition_functions",
    "robo.acquisition_functions.pi": "robo_amd.acquisition_functions",
    "robo.acquisition_functions.lcb": "robo_amd.acquisition_functions",
    "robo.acquisition_functions.information_gain": "robo_amd.acquisition_functions",
    "robo.acquisition_functions.information_gain_per_unit_cost": "robo_amd.acquisition_functions",
    "robo.acquisition_functions.marginalization": "robo_amd.acquisition_functions",
    "robo.acquisition_functions.base_acquisition": "robo_amd.acquisition_functions.base_acquisition",
    "robo.maximizers.random_sampling": "robo_amd.maximizers", "robo.maximizers.scipy_optimizer": "robo_amd.maximizers",
    "robo.maximizers.differential_evolution": "robo_amd.maximizers",
    "robo.maximizers.base_maximizer": "robo_amd.maximizers.random_sampling",
    "robo.maximizers.grid_search": "robo_amd.maximizers",
    "robo.solver.bayesian_optimization": "robo_amd.solver", "robo.solver.base_solver": "robo_amd.solver",
    "robo.priors.default_priors": "robo_amd.priors", "robo.priors.env_priors": "robo_amd.priors",
    "robo.priors.base_prior": "robo_amd.priors",
    "robo.fmin.bayesian_optimization": "robo_amd.fmin", "robo.fmin.entropy_search": "robo_amd.fmin",
    "robo.fmin.fabolas": "robo_amd.fmin", "robo.fmin.random_search": "robo_amd.fmin",
    "robo.initial_design.init_random_uniform": "robo_amd.initial_design",
    "robo.initial_design.init_latin_hypercube_sampling": "robo_amd.initial_design",
    "robo.initial_design.init_grid": "robo_amd.initial_design",
    "robo.initial_design.init_random_normal": "robo_amd.initial_design",
    "robo.util.incumbent_estimation": "robo_amd.util.incumbent_estimation",
    "robo.util.normalization": "robo_amd.util.normalization", "robo.util.epmgp": "robo_amd.util.epmgp",
    "robo.util.mc_part": "robo_amd.util.mc_part",
}

# stated differences (everything else must match)
ALLOWED = {
    # the reference wraps predict in BaseModel._check_shapes_predict, whose wrapper names the first argument ``X``
    # (base_model.py:31-44); the wrapped method's own name is X_test, which is what robo_amd exposes.  Positional either way.
    ("GaussianProcess", "predict"), ("GaussianProcessMCMC", "predict"), ("FabolasGPMCMC", "predict"),
}

with open(os.path.join(HERE, "golden", "ref_api_surface.json")) as _fh:
    SURFACE = json.load(_fh)


def _mine(sig_of):
    out = []
    for p in inspect.signature(sig_of).parameters.values():
        if p.default is inspect.Parameter.empty:
            d = None
        elif callable(p.default):
            d = "<callable>:" + getattr(p.default, "__name__", "?")
        else:
            d = repr(p.default)
        out.append([p.name, p.kind.name, d])
    return out


@pytest.mark.parametrize("key", sorted(k for k in SURFACE if k != "attributes"))
def test_same_call_surface(key):
    mod, name = key.split(":")
    obj = getattr(importlib.import_module(HOME[mod]), name)
    problems = []
    for method, ref_sig in SURFACE[key].items():
        if ref_sig is None:
            continue
        target = obj if method == "" else getattr(obj, method, None)
        if target is None:
            problems.append("%s.%s is missing" % (name, method))
            continue
        if (name, method) in ALLOWED:
            continue
        mine = _mine(target)
        lead = [p for p in ref_sig if not p[1].startswith("VAR_")]
        for i, (pname, _, default) in enumerate(lead):
            if i >= len(mine) or mine[i][0] != pname:
                problems.append("%s.%s: parameter %d is %r, the reference's is %r" %
                                (name, method, i, mine[i][0] if i < len(mine) else None, pname))
                break
            if mine[i][2] != default:
                problems.append("%s.%s(%s=...): default %s, the reference's %s" % (name, method, pname, mine[i][2], default))
        # whatever robo_amd adds behind the reference's parameters must be optional
        for p in mine[len(lead):]:
            if p[2] is None and not p[1].startswith("VAR_"):
                problems.append("%s.%s: extra parameter %s has no default" % (name, method, p[0]))
    assert not problems, "\n".join(problems)


def test_same_public_attributes():
    """after construction + train / update every object carries (at least) the public attributes the reference's carries
    (``model.X``, ``model.hypers``, ``model.models``, ``acq.estimators``, ``solver.incumbents`` ...): code that reads them
    keeps working.  robo_amd's objects are built through the interpreter (host logic only)."""
    import sys
    import types
    import numpy as np
    sys.path.insert(0, os.path.join(HERE, "hipemu"))
    import build_emu
    import make_golden_ref as G
    from robo_amd import _lib
    from robo_amd import acquisition_functions as A, maximizers as MX, models as M, priors as P
    from robo_amd.kernels import Matern52Kernel
    from robo_amd.solver import BayesianOptimization
    ns = types.SimpleNamespace(BayesianOptimization=BayesianOptimization)
    for mod in (A, MX, M, P):
        for k in dir(mod):
            if not k.startswith("_"):
                setattr(ns, k, getattr(mod, k))
    _lib.use_library(build_emu.build())
    try:
        lo, hi, X, y = G.api_attribute_data()
        objs = G.api_attribute_objects(ns, lambda: 2 * Matern52Kernel(np.ones(2), ndim=2), lo, hi, X, y)
        problems = []
        for name, want in SURFACE["attributes"].items():
            have = {a for a in vars(objs[name]) if not a.startswith("_")}
            missing = sorted(set(want) - have)
            if missing:
                problems.append("%s lacks %s" % (name, missing))
        assert not problems, "\n".join(problems)
    finally:
        _lib.use_library(None)

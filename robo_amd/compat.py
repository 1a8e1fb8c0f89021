"""Run code written against ``robo`` (and ``george.kernels``) on robo_amd without touching its imports.

    import robo_amd.compat; robo_amd.compat.install()
    from robo.fmin import bayesian_optimization            # robo_amd.fmin.bayesian_optimization
    import george; k = 2 * george.kernels.Matern52Kernel(np.ones(D), ndim=D)    # robo_amd.kernels

``robo`` and every ``robo.x.y`` resolve to the robo_amd module of the same path -- the SAME module objects, nothing is
copied -- for the part of the reference this package rebuilds (GP models, closed-form and information
-gain acquisition functions, the marginalisation, the three maximisers, the solver, priors, initial designs, utilities,
the three front ends).  ``george`` is a module with one attribute, ``kernels`` (the kernel descriptions of
robo_amd/kernels.py: Matern-5/2, squared-exponential, the Fabolas product; with and without the amplitude factor).
Anything else (``robo.models.random_forest``, ``george.GP`` ...) raises the ordinary ImportError / AttributeError.

The reference's own unit tests run this way (tools/run_reference_tests.py, tests/test_reference_suite.py).  It is an
import alias for callers, not a backend switch: with ``robo`` really installed, call ``install(force=True)`` to shadow it.
"""
import importlib
import importlib.abc
import importlib.util
import sys
import types

_installed = None


class _RoboAlias(importlib.abc.MetaPathFinder, importlib.abc.Loader):

    def find_spec(self, name, path=None, target=None):
        if name != "robo" and not name.startswith("robo."):
            return None
        try:
            if importlib.util.find_spec("robo_amd" + name[len("robo"):]) is None:
                return None
        except ModuleNotFoundError:
            return None
        return importlib.util.spec_from_loader(name, self)

    def create_module(self, spec):
        module = importlib.import_module("robo_amd" + spec.name[len("robo"):])
        # importlib's _init_module_attrs overwrites __spec__ / __loader__ of whatever create_module returns with the
        # ALIAS spec; the robo_amd module is live and must keep its own (reload, runpy, pkgutil and inspect read them):
        # remembered here, put back in exec_module
        self._own[id(module)] = (getattr(module, "__spec__", None), getattr(module, "__loader__", None),
                                 getattr(module, "__package__", None), getattr(module, "__path__", None),
                                 getattr(module, "__file__", None), getattr(module, "__cached__", None))
        return module

    def exec_module(self, module):
        own = self._own.pop(id(module), None)
        if own is None:
            return
        for attr, value in zip(("__spec__", "__loader__", "__package__", "__path__", "__file__", "__cached__"), own):
            if value is not None:
                setattr(module, attr, value)

    _own = {}


def install(force=False):
    """idempotent; ``force``: shadow an importable ``robo`` / ``george`` instead of refusing"""
    global _installed
    if _installed is not None:
        return
    for real in ("robo", "george"):
        present = real in sys.modules
        if not present:
            try:
                present = importlib.util.find_spec(real) is not None
            except (ImportError, ValueError):
                present = False
        if present and not force:
            raise RuntimeError("'%s' is importable here: robo_amd.compat.install(force=True) shadows it" % real)
    for name in [m for m in sys.modules if m in ("robo", "george") or m.startswith(("robo.", "george."))]:
        del sys.modules[name]
    import robo_amd.kernels as kernels
    george = types.ModuleType("george")
    george.__doc__ = "robo_amd.compat: the slice of george that RoBO callers use (george.kernels)"
    george.kernels = kernels
    sys.modules["george"], sys.modules["george.kernels"] = george, kernels
    _installed = _RoboAlias()
    sys.meta_path.insert(0, _installed)


def uninstall():
    global _installed
    if _installed is None:
        return
    sys.meta_path.remove(_installed)
    _installed = None
    for name in [m for m in sys.modules if m in ("robo", "george") or m.startswith(("robo.", "george."))]:
        del sys.modules[name]

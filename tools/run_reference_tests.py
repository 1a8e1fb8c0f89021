"""Run the REFERENCE'S OWN unit tests (/root/reference/test, unchanged, read-only) against robo_amd.

    python tools/run_reference_tests.py [--emu] [pattern ...]

``robo`` and every ``robo.x.y`` resolve to the robo_amd module of the same path (the package mirrors the reference's
module layout for everything on or next to the hot path); ``george.kernels`` resolves to robo_amd.kernels (the kernel
objects the reference's tests build and hand to the models).  Nothing of the reference's package is imported -- only its
test files run.  With --emu the library is the g++ interpreter build (no GPU needed; tests/hipemu), otherwise
librobo_hip.so on the GPU.  Needs /root/reference, so this runs in the build container only; tests/test_reference_suite.py
wraps it.

Files that test components SURVEY.md section 8 puts out of scope are not collected: random forest, Bohamiann, Bayesian
linear regression (models), GridSearch inside test_maximizers_* (the two files import it at module level).
"""
import importlib
import importlib.abc
import importlib.util
import os
import sys
import types
import unittest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_TESTS = "/root/reference/test"
IN_SCOPE = [
    "test_acquisition_functions/test_ei.py", "test_acquisition_functions/test_log_ei.py",
    "test_acquisition_functions/test_pi.py", "test_acquisition_functions/test_lcb.py",
    "test_acquisition_functions/test_marginalization.py", "test_acquisition_functions/test_information_gain.py",
    "test_acquisition_functions/test_information_gain_per_unit_cost.py",
    "test_models/test_gaussian_process.py", "test_models/test_gaussian_process_mcmc.py",
    "test_solver/test_bayesian_optimization.py", "test_initial_design/test_initial_design.py",
    "test_util/test_incumbent_estimation.py", "test_util/test_mc_part.py", "test_util/test_normalization.py",
    "test_util/test_posterior_optimization.py", "test_fmin/test_fabolas.py", "test_fmin/test_fmin_interface.py",
    "test_maximizer/test_maximizers_one_dim.py", "test_maximizer/test_maximizers_two_dim.py",
]


class _Alias(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    """robo[.x.y] -> robo_amd[.x.y] (the same module objects)"""

    def find_spec(self, name, path=None, target=None):
        if name == "robo" or name.startswith("robo."):
            return importlib.util.spec_from_loader(name, self)
        return None

    def create_module(self, spec):
        return importlib.import_module("robo_amd" + spec.name[len("robo"):])

    def exec_module(self, module):
        pass


def install_aliases():
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    sys.meta_path.insert(0, _Alias())
    import robo_amd.kernels as K
    george = types.ModuleType("george")
    george.kernels = K
    sys.modules["george"], sys.modules["george.kernels"] = george, K
    # out-of-scope optional back ends some test modules import at the top
    for mod, names in (("robo.maximizers.grid_search", ("GridSearch",)), ("robo.fmin.random_search", ())):
        m = types.ModuleType(mod)
        for n in names:
            setattr(m, n, None)
        sys.modules[mod] = m
    import robo_amd.fmin as F
    if not hasattr(F, "random_search"):
        F.random_search = None
    for attr, val in (("Infinity", float("inf")), ("NAN", float("nan"))):      # NumPy-1 names the tests may use
        import numpy as np
        if not hasattr(np, attr):
            setattr(np, attr, val)
    sys.path.insert(0, os.path.dirname(REF_TESTS))          # ``from test.dummy_model import DemoModel``


def load(patterns):
    suite = unittest.TestSuite()
    loader = unittest.TestLoader()
    for rel in IN_SCOPE:
        if patterns and not any(p in rel for p in patterns):
            continue
        name = "test." + rel[:-3].replace("/", ".")
        suite.addTests(loader.loadTestsFromName(name))
    return suite


def main(argv):
    emu = "--emu" in argv
    patterns = [a for a in argv if not a.startswith("--")]
    install_aliases()
    if emu:
        sys.path.insert(0, os.path.join(ROOT, "tests", "hipemu"))
        import build_emu
        from robo_amd import _lib
        _lib.use_library(build_emu.build())
    result = unittest.TextTestRunner(verbosity=1, stream=sys.stdout).run(load(patterns))
    print("REFERENCE-SUITE ran=%d failures=%d errors=%d skipped=%d" % (result.testsRun, len(result.failures),
                                                                      len(result.errors), len(result.skipped)))
    for kind, items in (("FAIL", result.failures), ("ERROR", result.errors)):
        for test, _ in items:
            print(kind, test.id())
    return 0 if result.wasSuccessful() else 1


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))

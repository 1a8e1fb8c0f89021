"""Run the REFERENCE'S OWN unit tests (/root/reference/test, unchanged, read-only) against robo_amd.

    python tools/run_reference_tests.py [--emu] [pattern ...]

``robo`` and every ``robo.x.y`` resolve to the robo_amd module of the same path (the package mirrors the reference's
module layout for everything on or next to the hot path); ``george.kernels`` resolves to robo_amd.kernels (the kernel
objects the reference's tests build and hand to the models).  Nothing of the reference's package is imported -- only its
test files run.  With --emu the library is the g++ interpreter build (no GPU needed; tests/hipemu), otherwise
librobo_hip.so on the GPU.  Needs /root/reference, so this runs in the build container only; tests/test_reference_suite.py
wraps it.

Files that test components SURVEY.md section 8 puts out of scope are not collected: random forest, Bohamiann, Bayesian
linear regression (models).
"""
import os
import sys
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


def install_aliases():
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    import robo_amd.compat
    robo_amd.compat.install(force=True)            # robo[.x.y] -> robo_amd[.x.y], george.kernels -> robo_amd.kernels
    import numpy as np
    for attr, val in (("Infinity", float("inf")), ("NAN", float("nan"))):      # NumPy-1 names the tests may use
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

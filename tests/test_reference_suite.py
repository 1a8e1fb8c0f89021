"""The REFERENCE'S OWN unit tests (/root/reference/test, unchanged) run against robo_amd: ``robo.x.y`` resolves to the
robo_amd module of the same path, ``george.kernels`` to robo_amd.kernels (tools/run_reference_tests.py), the library is the
interpreter build.  Build container only (the reference tree is not on the GPU box; there, run the tool without --emu by
hand if the tree is present).

52 tests in the 19 files on or next to the hot path.  Expected not to pass, and nothing else:
  * three front-end tests of out-of-scope model back ends (bohamiann, dngo, rf; SURVEY.md section 2 rows 6, 8: out of
    scope);
  * test_information_gain.test_innovations hands 1-D points to a 2-D model with unseeded random data: it fails on the
    reference itself as well (checked with the reference's own classes); it may pass or fail here.
"""
import os
import re
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)

pytestmark = pytest.mark.skipif(not os.path.isdir("/root/reference/test"), reason="reference tree not on this box")

OUT_OF_SCOPE = {
    "test.test_fmin.test_fmin_interface.TestFminInterface.test_bohamiann",
    "test.test_fmin.test_fmin_interface.TestFminInterface.test_dngo",
    "test.test_fmin.test_fmin_interface.TestFminInterface.test_rf",
}
BROKEN_IN_THE_REFERENCE = {"test.test_acquisition_functions.test_information_gain.TestInformationGain.test_innovations"}


def _run(patterns):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "run_reference_tests.py"), "--emu"] + patterns,
                       cwd=ROOT, capture_output=True, text=True, timeout=1500)
    m = re.search(r"REFERENCE-SUITE ran=(\d+) failures=(\d+) errors=(\d+)", r.stdout)
    assert m, r.stdout[-2000:] + r.stderr[-2000:]
    bad = set(re.findall(r"^(?:FAIL|ERROR) (\S+)$", r.stdout, flags=re.M))
    return int(m.group(1)), bad, r.stdout


def test_reference_unit_tests_pass_against_robo_amd():
    """everything but the Fabolas front-end test (next test): 51 tests, 47 must pass, 3 are out of scope, 1 is broken"""
    files = ["test_acquisition_functions", "test_models", "test_solver", "test_initial_design", "test_util",
             "test_maximizer", "test_fmin_interface"]
    ran, bad, out = _run(files)
    assert ran == 51, out[-1500:]
    assert OUT_OF_SCOPE <= bad, "an out-of-scope component started to import?"
    assert bad - OUT_OF_SCOPE - BROKEN_IN_THE_REFERENCE == set(), out[-3000:]


def test_reference_fabolas_front_end_test_passes():
    """test/test_fmin/test_fabolas.py: robo.fmin.fabolas end to end on the reference's own toy objective"""
    ran, bad, out = _run(["test_fabolas"])
    assert ran == 1 and not bad, out[-3000:]

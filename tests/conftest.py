import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.hookimpl(tryfirst=True)
def pytest_cmdline_main(config):
    """The CPU suite (`-m "not gpu"` on a box without a GPU) is ~9 minutes of single-threaded interpreter work in twelve
    independent files: run the files on four to six worker processes (pytest-xdist, --dist loadfile: a file's tests stay together
    and in order, module fixtures are per worker) unless the caller chose a process count, asked for a serial run
    (ROBO_TESTS_SERIAL=1) or xdist is absent.  NEVER on a GPU box: the `-m gpu` tests share one device and time things."""
    try:
        opt = config.option
        if os.environ.get("ROBO_TESTS_SERIAL") == "1" or os.environ.get("PYTEST_XDIST_WORKER"):
            return None
        if not _no_gpu_hardware() or getattr(opt, "markexpr", "") != "not gpu":
            return None
        if not config.pluginmanager.hasplugin("xdist") or getattr(opt, "numprocesses", None) is not None:
            return None
        if getattr(opt, "collectonly", False) or getattr(opt, "usepdb", False) or (os.cpu_count() or 1) < 4:
            return None
        opt.numprocesses = max(4, min(6, (os.cpu_count() or 4) - 2))
        opt.dist = "loadfile"
    except Exception:        # noqa: BLE001 -- any surprise: the plain serial run
        pass
    return None


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


def _no_gpu_hardware():
    """True only on a box without an AMD GPU device node.  On a GPU box nothing is ever skipped: a missing
    or broken librobo_hip.so must fail the gpu tests loudly, not turn them into skips."""
    return not os.path.exists("/dev/kfd")


def pytest_collection_modifyitems(config, items):
    if not _no_gpu_hardware():
        return
    skip = pytest.mark.skip(reason="no AMD GPU on this box (/dev/kfd absent); gpu-marked tests run with -m gpu on the MI355X")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)

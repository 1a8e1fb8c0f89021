"""The C-ABI library loads and exports every symbol include/robo_hip.h declares (no compute
calls: there is no GPU in the build container), and the product refuses to run without it."""
import ctypes
import os
import re

import pytest

from robo_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header="robo_hip.h"):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(robo_[a-z0-9_]+)\s*\(", text)))


def test_binding_lists_every_declared_symbol():
    assert _declared() == sorted(_lib.SYMBOLS)
    assert _declared("robo_hip_diag.h") == sorted(_lib.DIAG_SYMBOLS)


def test_library_exports_every_declared_symbol():
    if not os.path.exists(_lib.DEFAULT_LIBRARY) or not os.path.exists(_lib.DEFAULT_DIAG_LIBRARY):
        from robo_amd import build
        build.build(verbose=False)
    handle = ctypes.CDLL(_lib.DEFAULT_LIBRARY)
    missing = [s for s in _declared() if not hasattr(handle, s)]
    assert not missing, missing
    # measurement code is not in the product library
    assert not [s for s in _lib.DIAG_SYMBOLS if hasattr(handle, s)]
    diag = ctypes.CDLL(_lib.DEFAULT_DIAG_LIBRARY)
    assert not [s for s in _declared("robo_hip_diag.h") if not hasattr(diag, s)]


def test_no_torch_types_in_the_abi():
    text = open(os.path.join(ROOT, "include", "robo_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)      # declarations only
    assert "torch" not in text.lower() and "at::" not in text and "#include <hip" not in text


def test_product_fails_loudly_without_the_extension(tmp_path):
    _lib.use_library(str(tmp_path / "nope.so"))
    try:
        with pytest.raises(_lib.RoboHipUnavailable):
            _lib.lib()
    finally:
        _lib.use_library(None)


def test_product_never_imports_the_oracle():
    """only tests/, smoke() and bench.py's cpu_baseline may touch oracle/"""
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, "robo_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f), errors="replace").read()
                if re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M) or "import gp_oracle" in src:
                    bad.append(os.path.join(dirpath, f))
    assert not bad, bad

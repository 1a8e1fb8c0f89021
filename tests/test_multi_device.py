"""Single-process multi-device entry points (robo_amd/csrc/multi.hip, include/robo_hip.h "multi-GPU, ONE process")
through the g++ interpreter build with HIPEMU_DEVICES emulated devices: 2 and 3 devices in ONE process (and two contexts
on one device) must give the single-device result for the candidate shard, the sample shard, the per-unit-cost shard,
the replicated / batched fits and the mixture posterior.  CPU-only; the GPU suite repeats the checks with two contexts
on the MI355X (tests/test_gpu_parity.py::test_multi_device_*)."""
import os
import sys

import numpy as np
import pytest

import multi_checks as MC
from robo_amd import _lib

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def emu3():
    sys.path.insert(0, os.path.join(HERE, "hipemu"))
    import build_emu
    path = build_emu.build()
    old = os.environ.get("HIPEMU_DEVICES")
    os.environ["HIPEMU_DEVICES"] = "3"
    _lib.use_library(path)
    assert _lib.device_count() == 3
    yield
    _lib.use_library(None)
    if old is None:
        del os.environ["HIPEMU_DEVICES"]
    else:
        os.environ["HIPEMU_DEVICES"] = old


@pytest.mark.parametrize("devices", [[0, 1], [0, 1, 2], [0, 0]])
def test_candidate_shard(emu3, devices):
    MC.check_candidate_shard(devices)


@pytest.mark.parametrize("devices", [[0, 1], [0, 1, 2]])
def test_sample_shard(emu3, devices):
    MC.check_sample_shard(devices)


@pytest.mark.parametrize("devices", [[0, 1], [0, 1, 2]])
def test_per_unit_cost_shard(emu3, devices):
    MC.check_per_cost_shard(devices)


def test_fits_and_mixture(emu3):
    MC.check_fits_and_mixture([0, 1, 2])


def test_failing_device_is_reported_not_hung(emu3):
    MC.check_failing_device([0, 1])


def test_same_results_without_worker_threads(emu3, monkeypatch):
    monkeypatch.setenv("ROBO_MULTI_THREADS", "0")
    _lib._multis.clear()
    try:
        assert _lib.multi_for([0, 1]).info()[2] == 0
        MC.check_candidate_shard([0, 1])
        MC.check_sample_shard([0, 1])
    finally:
        _lib._multis.clear()


def test_device_resolution(emu3):
    assert _lib.resolve_devices(None, None) is None
    assert _lib.resolve_devices(None, 1) is None
    assert _lib.resolve_devices(None, 3) == [0, 1, 2]
    assert _lib.resolve_devices([2, 0]) == [2, 0]
    with pytest.raises(ValueError):
        _lib.resolve_devices([0, 5])
    m = _lib.multi_for([0, 1, 2])
    assert m.info()[:2] == (3, [0, 1, 2]) and m is _lib.multi_for((0, 1, 2))
    assert [_lib.shard_range(50, g, 4) for g in range(4)] == [(0, 13), (13, 26), (26, 38), (38, 50)]

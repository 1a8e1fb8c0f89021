"""The interpreter's DEFERRED stream schedules (tests/hipemu/hipemu.cpp, HIPEMU_ASYNC / hipemu_set_async).

The default schedule executes every launch when it is issued, so a missing hipStreamWaitEvent between two streams can
never show on the CPU -- the round-5 fork race of the batched factorisation (the batch's gram launch issued behind the
fork event) was found only on the MI355X.  The two deferred schedules are legal orders of the same work that put the
OTHER streams first (1) or last (2) at every synchronisation point; the multi-stream paths of the library must give the
same bits under all three.  Host logic only: nothing here is a claim about the hardware.
"""
import ctypes
import os
import sys

import numpy as np
import pytest

from robo_amd import _lib

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "hipemu"))
sys.path.insert(0, os.path.join(HERE, "golden"))


def test_schedules_expose_a_known_race():
    """a fork/join with one dependency left out: hidden by schedule 0, exposed by exactly the schedule built for it"""
    import build_emu
    lib = ctypes.CDLL(build_emu.build_sched_selftest())
    lib.sched_selftest.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    out = np.zeros(64)

    def run(mode, fork, join):
        out[:] = -1.0
        assert lib.sched_selftest(mode, fork, join, out.ctypes.data) == 0
        assert np.all(out == out[0])
        return float(out[0])

    for mode in (0, 1, 2):
        assert run(mode, 1, 1) == 1.0                    # correct program: every schedule
    assert run(0, 0, 1) == 1.0 and run(0, 1, 0) == 1.0   # issue order hides both bugs
    assert run(1, 0, 1) == 0.0                           # side stream ran before its producer on the main stream
    assert run(2, 1, 0) == 0.0                           # main stream read the side stream's output before it existed
    assert run(2, 0, 1) == 1.0 and run(1, 1, 0) == 1.0   # (each schedule is blind to the other's bug)


@pytest.fixture(scope="module")
def emu():
    import build_emu
    path = build_emu.build()
    _lib.use_library(path)
    handle = ctypes.CDLL(path)
    handle.hipemu_deferred_total.restype = ctypes.c_long
    yield handle
    handle.hipemu_set_async(0)
    _lib.use_library(None)


@pytest.mark.parametrize("mode", [1, 2])
def test_multi_stream_paths_under_deferred_schedules(emu, mode, monkeypatch):
    """the batched factorisation's sub-batch streams (fork, staggered groups, join), the sample shard's peer copies and the
    candidate shard over two emulated devices: same results whichever legal order the streams' work runs in"""
    monkeypatch.setenv("HIPEMU_DEVICES", "2")
    import multi_checks as M
    import parity_checks as P
    before = emu.hipemu_deferred_total()
    emu.hipemu_set_async(mode)
    try:
        ctx = _lib.Context(0)
        P.check_batched_split(ctx, N=300, variants=((2, 2, -1),))
        M.check_sample_shard([0, 1])
        M.check_candidate_shard([0, 1])
    finally:
        emu.hipemu_set_async(0)
    assert emu.hipemu_deferred_total() > before + 100    # the work really went through the stream queues


def _child(code_or_args, env_extra, timeout=900):
    import subprocess
    env = dict(os.environ, **env_extra)
    env["PYTHONPATH"] = os.pathsep.join([os.path.dirname(HERE), env.get("PYTHONPATH", "")])
    return subprocess.run([sys.executable] + code_or_args, env=env, cwd=os.path.dirname(HERE), capture_output=True,
                          text=True, timeout=timeout)


def test_guard_pages_fault_on_an_overrun():
    """HIPEMU_GUARD=1: one element past the end of a device buffer is a fault at the store, not a silent neighbour write"""
    import build_emu
    lib = build_emu.build_sched_selftest()
    probe = "import ctypes,sys; sys.exit(ctypes.CDLL(%r).guard_probe(int(sys.argv[1])))" % lib
    assert _child(["-c", probe, "0"], {"HIPEMU_GUARD": "1"}).returncode == 0
    assert _child(["-c", probe, "1"], {"HIPEMU_GUARD": "1"}).returncode == -11       # SIGSEGV
    assert _child(["-c", probe, "1"], {"HIPEMU_GUARD": "0"}).returncode == 0         # (unnoticed without the guard)


_FAST_SUBSET = ("edge_sizes or shape_sweep or golden_cases or multi_panel or chunked_workspace or mcmc_marginal or "
                "batched_likelihoods or fit_batch_keeps or ill_conditioned or fabolas_kernel or candidate_reupload or "
                "argmax_semantics or model_gradients")


@pytest.mark.parametrize("env", [{"HIPEMU_GUARD": "1", "HIPEMU_ORDER": "1"}, {"HIPEMU_ORDER": "2"}],
                         ids=["fenced+descending", "rotating"])
def test_kernels_stay_inside_their_buffers_in_any_work_item_order(env):
    """a broad, fast subset of the interpreter tests again (ragged and edge sizes, multi-panel factorisation, chunked
    workspaces, batched likelihoods, the mixture and the argmax paths): once with every device buffer fenced by
    inaccessible pages and the work-items of a workgroup run in descending order, once in rotating order"""
    r = _child(["-m", "pytest", os.path.join(HERE, "test_emu_logic.py"), "-x", "-q", "-p", "no:cacheprovider",
                "-k", _FAST_SUBSET], dict(env, ROBO_TESTS_SERIAL="1"))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and "failed" not in r.stdout


def test_work_item_order_exposes_a_missing_barrier():
    """HIPEMU_ORDER=1 (descending) / 2 (rotating): a wave that reads LDS another wave wrote, without __syncthreads in
    between, sees stale data; the default ascending order hides it.  With the barrier every order agrees."""
    import build_emu
    lib = build_emu.build_sched_selftest()
    probe = "import ctypes,sys; sys.exit(ctypes.CDLL(%r).order_probe(int(sys.argv[1])))" % lib
    for order in ("0", "1", "2"):
        assert _child(["-c", probe, "1"], {"HIPEMU_ORDER": order}).returncode == 0
    # ascending order hides it (all but the work-item that arrived last at the previous barrier and runs on first)
    assert _child(["-c", probe, "0"], {"HIPEMU_ORDER": "0"}).returncode <= 1
    assert _child(["-c", probe, "0"], {"HIPEMU_ORDER": "1"}).returncode >= 63         # (nearly) every element stale

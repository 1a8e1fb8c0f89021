"""ctypes binding of librobo_hip.so -- the C ABI declared in include/robo_hip.h.

This is the binding a RoBO maintainer would add.  There is no CPU
implementation behind it: if the shared library is missing or no HIP device is visible the
first use raises :class:`RoboHipUnavailable`.

``use_library(path)`` lets the test-suite point the binding at the g++-interpreted build of
the same sources (tests/hipemu) to exercise host logic in the GPU-less build container; the
product never calls it.
"""
import atexit
import ctypes as C
import os
import weakref

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIBRARY = os.path.join(_HERE, "librobo_hip.so")
DEFAULT_DIAG_LIBRARY = os.path.join(_HERE, "librobo_hip_diag.so")   # self-checks / micro-benchmarks (not product)

OK, NOT_POSITIVE_DEFINITE, NOT_FITTED, BAD_SHAPE, RUNTIME_ERROR, BAD_ARGUMENT = range(6)
KERNEL_KINDS = {"matern52": 0, "rbf": 1, "fabolas": 2}
ACQ_KINDS = {"ei": 0, "log_ei": 1, "pi": 2, "lcb": 3}
FLAG_ZERO_SIGMA, FLAG_NEGATIVE_EI, FLAG_NAN = 1, 2, 4

# every symbol include/robo_hip.h declares (tests check the library exports all of them)
SYMBOLS = [
    "robo_device_count", "robo_ctx_create", "robo_ctx_destroy", "robo_ctx_live_count", "robo_ctx_synchronize",
    "robo_ctx_device_name", "robo_ctx_event_record", "robo_ctx_event_elapsed_ms", "robo_ctx_set_phase_events",
    "robo_ctx_set_tuning",
    "robo_last_error_string", "robo_version_string",
    "robo_gp_create", "robo_gp_destroy", "robo_gp_set_data", "robo_gp_set_output_transform",
    "robo_gp_set_precision", "robo_theta_size",
    "robo_gp_fit", "robo_gp_loglik_batch", "robo_gp_mcmc_run", "robo_mcmc_draws", "robo_gp_fit_batch", "robo_gp_grad_loglik", "robo_gp_get_factor", "robo_gp_get_gram", "robo_gp_factor_cond", "robo_gp_prefetch_inverse",
    "robo_cand_create", "robo_cand_destroy", "robo_cand_set_points", "robo_cand_create_uniform", "robo_cand_get_points",
    "robo_cand_create_random", "robo_cand_create_sobol", "robo_cand_get_point", "robo_cand_workspace_chunk", "robo_cand_last_solve_kernel",
    "robo_gp_predict_cand", "robo_gp_predict", "robo_gp_predict_cov", "robo_gp_predict_grad", "robo_gp_predict_mixture_cand",
    "robo_acq_eval_cand", "robo_acq_eval", "robo_acq_eval_moments", "robo_acq_eval_marginal_cand", "robo_acq_eval_sum_cand",
    "robo_ig_eval_cand", "robo_ig_eval_per_cost_cand", "robo_ig_eval_moments", "robo_gp_cross_cov",
    "robo_comm_create_id", "robo_comm_init", "robo_comm_destroy", "robo_comm_info", "robo_comm_allgather",
    "robo_acq_eval_cand_sharded", "robo_acq_eval_marginal_cand_sharded", "robo_ig_eval_per_cost_cand_sharded",
    "robo_multi_create", "robo_multi_destroy", "robo_multi_info", "robo_gp_set_data_multi", "robo_gp_fit_multi",
    "robo_gp_loglik_batch_multi", "robo_gp_fit_batch_multi", "robo_acq_eval_cand_multi",
    "robo_ig_eval_cand_multi", "robo_ig_eval_per_cost_cand_multi", "robo_acq_eval_marginal_cand_multi",
    "robo_gp_predict_mixture_cand_multi",
]
COMM_ID_BYTES = 128
# include/robo_hip_diag.h (librobo_hip_diag.so: tests, bench.py's roofline block, tools/)
DIAG_SYMBOLS = [
    "robo_selftest_mfma_layout", "robo_microbench_mfma_f64", "robo_microbench_mfma_f64_detail", "robo_microbench_gemm_f64",
    "robo_selftest_diag_timeline", "robo_diag_clock_sample_begin", "robo_diag_clock_sample_end",
    "robo_selftest_stretch_move",
]


class RoboHipUnavailable(RuntimeError):
    pass


class RoboHipError(RuntimeError):
    pass


class RoboBadShape(AssertionError):
    """ROBO_BAD_SHAPE from the library (the reference uses ``assert`` for shapes, base_model.py:68-70,76) -- its own type
    so that callers can tell the library's verdict from a Python-side assert"""


_lib = None
_lib_path = None
_diag = None
_dp = C.POINTER(C.c_double)


def _arr(a):
    return a.ctypes.data_as(_dp)


def _f64(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None and a.shape != shape:
        raise AssertionError("expected shape %r, got %r" % (shape, a.shape))
    return a


def use_library(path):
    """Point the binding at another build of the same C ABI (test hook)."""
    global _lib, _lib_path, _default_ctx, _diag, _extra_ctx, _multis
    _close_multis()                        # worker threads of the outgoing library end here, not at garbage collection
    _lib = None
    _diag = None
    _lib_path = path
    _default_ctx = {}
    _extra_ctx = {}
    _multis = {}


def library_path():
    return _lib_path or DEFAULT_LIBRARY


def lib():
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path):
        raise RoboHipUnavailable(
            "%s not found: build it with `python -m robo_amd.build` (hipcc, gfx950). "
            "robo_amd has no CPU fallback." % path)
    try:
        L = C.CDLL(path)
    except OSError as e:
        raise RoboHipUnavailable("cannot load %s: %s" % (path, e))
    vp, i32, i64, dbl = C.c_void_p, C.c_int32, C.c_int64, C.c_double
    pp = C.POINTER(vp)
    sig = {
        "robo_device_count": [C.POINTER(i32)],
        "robo_ctx_create": [i32, vp, pp],
        "robo_ctx_destroy": [vp],
        "robo_ctx_live_count": [C.POINTER(i32)],
        "robo_ctx_synchronize": [vp],
        "robo_ctx_device_name": [vp, C.c_char_p, i32],
        "robo_ctx_event_record": [vp, i32],
        "robo_ctx_event_elapsed_ms": [vp, i32, i32, C.POINTER(C.c_float)],
        "robo_ctx_set_phase_events": [vp, i32],
        "robo_ctx_set_tuning": [vp, C.c_char_p, i64],
        "robo_gp_create": [vp, i32, i32, i32, pp],
        "robo_gp_destroy": [vp],
        "robo_gp_set_data": [vp, _dp, _dp, i32],
        "robo_gp_set_output_transform": [vp, dbl, dbl],
        "robo_gp_set_precision": [vp, i32],
        "robo_theta_size": [i32, i32],
        "robo_gp_fit": [vp, _dp, dbl, _dp, C.POINTER(i32)],
        "robo_gp_loglik_batch": [vp, _dp, i32, dbl, _dp, C.POINTER(i32)],
        "robo_gp_fit_batch": [pp, i32, _dp, dbl, _dp, C.POINTER(i32)],
        "robo_mcmc_draws": [C.POINTER(C.c_uint32), C.POINTER(i32), i32, i32, _dp, C.POINTER(i32), _dp],
        "robo_gp_mcmc_run": [vp, dbl, i32, _dp, i32, i32, dbl, _dp, C.POINTER(i32), _dp, i32, _dp, _dp, _dp, _dp,
                             C.POINTER(i64)],
        "robo_gp_grad_loglik": [vp, _dp, dbl, _dp, _dp, C.POINTER(i32)],
        "robo_gp_get_factor": [vp, _dp],
        "robo_gp_get_gram": [vp, _dp, _dp],
        "robo_gp_factor_cond": [vp, _dp],
        "robo_gp_prefetch_inverse": [vp],
        "robo_cand_create": [vp, _dp, i64, i32, pp],
        "robo_cand_destroy": [vp],
        "robo_cand_set_points": [vp, _dp, i64],
        "robo_cand_create_uniform": [vp, i64, i32, C.c_uint64, pp],
        "robo_cand_get_points": [vp, _dp],
        "robo_cand_create_random": [vp, i64, i32, C.c_uint64, i64, _dp, _dp, pp],
        "robo_cand_create_sobol": [vp, i64, i32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), i32, C.c_uint64, pp],
        "robo_cand_get_point": [vp, i64, _dp],
        "robo_cand_workspace_chunk": [vp, C.POINTER(i64)],
        "robo_cand_last_solve_kernel": [vp, C.c_char_p, i32],
        "robo_gp_predict_cand": [vp, vp, _dp, _dp],
        "robo_gp_predict": [vp, _dp, i64, _dp, _dp],
        "robo_gp_predict_cov": [vp, _dp, i64, _dp, _dp],
        "robo_gp_predict_grad": [vp, _dp, i64, _dp, _dp, _dp, _dp],
        "robo_gp_predict_mixture_cand": [pp, i32, vp, _dp, _dp],
        "robo_acq_eval_cand": [vp, i32, dbl, dbl, vp, _dp, _dp, C.POINTER(i64), C.POINTER(C.c_uint32)],
        "robo_acq_eval": [vp, i32, dbl, dbl, _dp, i64, _dp, _dp, C.POINTER(i64), C.POINTER(C.c_uint32)],
        "robo_acq_eval_moments": [vp, i32, dbl, dbl, _dp, _dp, i64, _dp, _dp, C.POINTER(i64), C.POINTER(C.c_uint32)],
        "robo_acq_eval_marginal_cand": [pp, i32, i32, dbl, _dp, vp, _dp, _dp, C.POINTER(i64),
                                        C.POINTER(C.c_uint32)],
        "robo_acq_eval_sum_cand": [pp, i32, i32, dbl, _dp, vp, _dp, C.POINTER(C.c_uint32)],
        "robo_ig_eval_cand": [vp, vp, vp, i32, dbl, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, C.POINTER(i64)],
        "robo_ig_eval_per_cost_cand": [vp, vp, vp, i32, dbl, _dp, _dp, _dp, _dp, _dp, _dp, vp, vp, dbl, _dp, _dp,
                                       C.POINTER(i64)],
        "robo_ig_eval_per_cost_cand_sharded": [vp, vp, vp, vp, i32, dbl, _dp, _dp, _dp, _dp, _dp, _dp, vp, vp, dbl, i64,
                                               _dp, _dp, C.POINTER(i64), C.POINTER(i32)],
        "robo_ig_eval_moments": [vp, i64, i32, i32, dbl, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp],
        "robo_gp_cross_cov": [vp, vp, vp, _dp],
        "robo_comm_create_id": [C.c_char_p],
        "robo_comm_init": [vp, i32, i32, C.c_char_p, pp],
        "robo_comm_destroy": [vp],
        "robo_comm_info": [vp, C.POINTER(i32), C.POINTER(i32)],
        "robo_comm_allgather": [vp, _dp, i64, _dp],
        "robo_acq_eval_cand_sharded": [vp, vp, i32, dbl, dbl, vp, i64, _dp, _dp, C.POINTER(i64), C.POINTER(i32),
                                       C.POINTER(C.c_uint32)],
        "robo_acq_eval_marginal_cand_sharded": [vp, pp, i32, i32, i32, dbl, _dp, vp, _dp, _dp, C.POINTER(i64),
                                                C.POINTER(C.c_uint32)],
        "robo_multi_create": [pp, i32, pp],
        "robo_multi_destroy": [vp],
        "robo_multi_info": [vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)],
        "robo_gp_set_data_multi": [vp, pp, _dp, _dp, i32],
        "robo_gp_fit_multi": [vp, pp, _dp, dbl, _dp, C.POINTER(i32)],
        "robo_gp_loglik_batch_multi": [vp, pp, _dp, i32, dbl, _dp, C.POINTER(i32)],
        "robo_gp_fit_batch_multi": [vp, pp, C.POINTER(i32), _dp, dbl, _dp, C.POINTER(i32)],
        "robo_acq_eval_cand_multi": [vp, pp, i32, dbl, dbl, pp, C.POINTER(i64), _dp, _dp, C.POINTER(i64), C.POINTER(i32),
                                     C.POINTER(C.c_uint32)],
        "robo_ig_eval_cand_multi": [vp, pp, pp, pp, i32, dbl, _dp, _dp, _dp, _dp, _dp, _dp, C.POINTER(i64), _dp, _dp,
                                    C.POINTER(i64), C.POINTER(i32)],
        "robo_ig_eval_per_cost_cand_multi": [vp, pp, pp, pp, i32, dbl, _dp, _dp, _dp, _dp, _dp, _dp, pp, pp, dbl,
                                             C.POINTER(i64), _dp, _dp, C.POINTER(i64), C.POINTER(i32)],
        "robo_acq_eval_marginal_cand_multi": [vp, pp, C.POINTER(i32), i32, dbl, _dp, pp, _dp, _dp, C.POINTER(i64),
                                              C.POINTER(C.c_uint32)],
        "robo_gp_predict_mixture_cand_multi": [vp, pp, C.POINTER(i32), pp, _dp, _dp],
    }
    for name, args in sig.items():
        fn = getattr(L, name)
        fn.argtypes = args
        fn.restype = i32
    L.robo_last_error_string.restype = C.c_char_p
    L.robo_last_error_string.argtypes = []
    L.robo_version_string.restype = C.c_char_p
    L.robo_version_string.argtypes = []
    n = i32(0)
    L.robo_device_count(C.byref(n))
    if n.value < 1:
        raise RoboHipUnavailable("librobo_hip loaded but no HIP device is visible (robo_amd has no CPU fallback)")
    _lib = L
    return L


def diag():
    """the diagnostics library (self-checks, micro-benchmarks); the interpreter build of the tests is monolithic"""
    global _diag
    if _diag is not None:
        return _diag
    lib()                                            # the product library first (the diagnostics link against it)
    path = _lib_path or DEFAULT_DIAG_LIBRARY
    if not os.path.exists(path):
        raise RoboHipUnavailable("%s not found: build it with `python -m robo_amd.build`" % path)
    D = C.CDLL(path)
    vp, i32 = C.c_void_p, C.c_int32
    for name, args in {"robo_selftest_mfma_layout": [vp, _dp], "robo_microbench_mfma_f64": [vp, i32, _dp],
                       "robo_microbench_mfma_f64_detail": [vp, i32, _dp],
                       "robo_microbench_gemm_f64": [vp, i32, i32, i32, i32, _dp],
                       "robo_selftest_diag_timeline": [vp, _dp, _dp],
                       "robo_diag_clock_sample_begin": [vp, i32], "robo_diag_clock_sample_end": [vp, _dp],
                       "robo_selftest_stretch_move": [vp, _dp, _dp, _dp, C.c_double, i32, i32, _dp, _dp, _dp]}.items():
        fn = getattr(D, name)
        fn.argtypes = args
        fn.restype = i32
    _diag = D
    return D


def last_error():
    return lib().robo_last_error_string().decode("utf-8", "replace")


def check(status, msg=None):
    """Map a robo_status to the exception the reference raises at the same point.  msg: the text to use instead of the
    library's last error string (a status that arrived from ANOTHER rank has no local error string)."""
    if status == OK:
        return
    msg = last_error() if msg is None else msg
    if status == NOT_POSITIVE_DEFINITE:
        raise np.linalg.LinAlgError(msg or "matrix is not positive definite")
    if status == NOT_FITTED:
        raise Exception('Model has to be trained first!')    # gaussian_process.py:241,273,322
    if status == BAD_SHAPE:
        raise RoboBadShape(msg)                               # base_model.py:68-70,76 use assert
    if status == BAD_ARGUMENT:
        raise ValueError(msg)
    raise RoboHipError(msg)


def live_contexts():
    """contexts whose device resources are still held by the library (robo_ctx_live_count)"""
    n = C.c_int32(0)
    lib().robo_ctx_live_count(C.byref(n))
    return n.value


def device_count():
    n = C.c_int32(0)
    lib().robo_device_count(C.byref(n))
    return n.value


def mcmc_draws(random_state, n_steps, half):
    """(u_stretch, partner, u_accept), each (n_steps, 2, half): the numbers `random_state` (a legacy numpy RandomState)
    would produce for n_steps emcee-2 ensemble steps -- rand(half), randint(half, size=half), rand(half) per half-step --
    drawn by the library (robo_mcmc_draws) and with the stream advanced exactly as those calls would have advanced it."""
    name, key, pos, has_gauss, cached = random_state.get_state()
    if name != "MT19937":
        raise ValueError("legacy MT19937 RandomState expected")
    key = np.ascontiguousarray(key, dtype=np.uint32).copy()
    p = C.c_int32(int(pos))
    uz = np.empty((n_steps, 2, half))
    ua = np.empty((n_steps, 2, half))
    pa = np.empty((n_steps, 2, half), dtype=np.int32)
    check(lib().robo_mcmc_draws(key.ctypes.data_as(C.POINTER(C.c_uint32)), C.byref(p), int(n_steps), int(half), _arr(uz),
                                pa.ctypes.data_as(C.POINTER(C.c_int32)), _arr(ua)))
    random_state.set_state((name, key, int(p.value), has_gauss, cached))
    return uz, pa, ua


class Context(object):
    """robo_ctx: one device, one HIP stream, 32 event slots."""

    def __init__(self, device=0, stream=None):
        self._h = C.c_void_p()
        self._owner = lib()          # the library that owns the handle (tests switch libraries: use_library)
        check(self._owner.robo_ctx_create(int(device), C.c_void_p(stream) if stream else None, C.byref(self._h)))
        self.device = int(device)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._owner.robo_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def synchronize(self):
        check(lib().robo_ctx_synchronize(self._h))

    @property
    def name(self):
        buf = C.create_string_buffer(256)
        check(lib().robo_ctx_device_name(self._h, buf, 256))
        return buf.value.decode()

    def record(self, slot):
        check(lib().robo_ctx_event_record(self._h, int(slot)))

    def elapsed_ms(self, a, b):
        ms = C.c_float(0)
        check(lib().robo_ctx_event_elapsed_ms(self._h, int(a), int(b), C.byref(ms)))
        return float(ms.value)

    def set_phase_events(self, on):
        """record event slots 19..23 around the phases of robo_gp_fit (off by default)"""
        check(lib().robo_ctx_set_phase_events(self._h, 1 if on else 0))

    def set_tuning(self, key, value=None):
        """kernel-variant / workspace knobs of this context (include/robo_hip.h); value None restores the default,
        key "env" re-reads the ROBO_* environment variables"""
        v = -(2 ** 63) if value is None else int(value)
        check(lib().robo_ctx_set_tuning(self._h, key.encode(), v))

    def selftest_mfma_layout(self):
        e = C.c_double(0)
        check(diag().robo_selftest_mfma_layout(self._h, C.byref(e)))
        return e.value

    def selftest_stretch_move(self, c, s, u, a=2.0, P=18):
        """-> (z, q, lnpdiff) of the chain's stretch-move device functions on arrays (include/robo_hip_diag.h)"""
        c, s, u = (np.ascontiguousarray(v, dtype=np.float64) for v in (c, s, u))
        assert c.shape == s.shape == u.shape and c.ndim == 1
        z, q, d = np.empty_like(c), np.empty_like(c), np.empty_like(c)
        check(diag().robo_selftest_stretch_move(self._h, _arr(c), _arr(s), _arr(u), float(a), int(P), c.size,
                                                _arr(z), _arr(q), _arr(d)))
        return z, q, d

    def microbench_mfma_f64_detail(self, iters=2000):
        """-> dict(full-chip tflops, issue interval of one lone wave in shader cycles, MHz under load)"""
        out = np.zeros(4)
        check(diag().robo_microbench_mfma_f64_detail(self._h, int(iters), _arr(out)))
        return {"tflops": out[0], "cycles_per_mfma_single_wave": out[1], "shader_mhz": out[2],
                "cycles_per_mfma_dependent_chain": out[3]}

    def microbench_gemm_f64(self, variant, wgs=512, k=4096, reps=5):
        out = np.zeros(2)
        check(diag().robo_microbench_gemm_f64(self._h, int(variant), int(wgs), int(k), int(reps), _arr(out)))
        return float(out[0]), float(out[1])    # TFLOP/s, shader MHz

    def clock_sample_begin(self, window_us):
        """Shader-clock sampler on a private stream (include/robo_hip_diag.h); measurement only."""
        check(diag().robo_diag_clock_sample_begin(self._h, int(window_us)))

    def clock_sample_end(self):
        out = np.zeros(3)
        check(diag().robo_diag_clock_sample_end(self._h, _arr(out)))
        return {"mean": float(out[0]), "min": float(out[1]), "max": float(out[2])}

    def microbench_mfma_f64(self, iters=2000):
        t = C.c_double(0)
        check(diag().robo_microbench_mfma_f64(self._h, int(iters), C.byref(t)))
        return t.value


_default_ctx = {}


def default_context(device=None):
    if device is None:
        device = int(os.environ.get("ROBO_DEVICE", os.environ.get("LOCAL_RANK", "0")))
        if device >= device_count():
            device = 0
    if device not in _default_ctx:
        _default_ctx[device] = Context(device)
    return _default_ctx[device]


_extra_ctx = {}        # (device, k): the k-th additional context on a device that appears more than once in a device list
_multis = {}           # tuple(devices) -> Multi
_live_multis = weakref.WeakSet()


def _close_multis():
    """End every Multi's worker threads while the interpreter and the HIP runtime are fully alive.  Registered with atexit:
    left to garbage collection during interpreter finalisation, the join of a worker thread (whose exit runs the HIP
    runtime's thread-local destructors) was seen to hang or abort the process on the MI355X (r05: a test process whose last
    test had used a device list)."""
    for m in list(_live_multis):
        try:
            m.close()
        except Exception:      # noqa: BLE001
            pass
    _multis.clear()


atexit.register(_close_multis)


def resolve_devices(devices=None, n_gpus=None):
    """the device list of a single-process multi-device run, or None for the ordinary one-device case:
    ``devices`` = explicit HIP device ids (a device may appear twice: two contexts on it, for tests on a one-GPU box),
    ``n_gpus`` = G -> devices 0 .. G-1"""
    if devices is None and n_gpus is not None and int(n_gpus) > 1:
        devices = list(range(int(n_gpus)))
    if devices is None:
        return None
    devices = [int(d) for d in devices]
    n = device_count()
    if any(d < 0 or d >= n for d in devices):
        raise ValueError("devices %r: this process sees %d HIP device(s)" % (devices, n))
    return devices if len(devices) > 1 else None


def contexts_for(devices):
    """one Context per entry of ``devices``: the process-wide default context of a device for its first occurrence, an
    extra one (kept for the life of the process) for every further occurrence"""
    seen = {}
    out = []
    for d in devices:
        k = seen.get(d, 0)
        seen[d] = k + 1
        if k == 0:
            out.append(default_context(d))
        else:
            if (d, k) not in _extra_ctx:
                _extra_ctx[(d, k)] = Context(d)
            out.append(_extra_ctx[(d, k)])
    return out


def multi_for(devices):
    """the process-wide Multi of a device list (every model / acquisition function given the same list shares it)"""
    key = tuple(int(d) for d in devices)
    if key not in _multis:
        _multis[key] = Multi(contexts_for(key))
    return _multis[key]


def shard_range(n_items, slot, n_slots):
    """contiguous [begin, end) of a slot's shard; the first n_items % n_slots slots hold one more (the library's rule)"""
    base, rem = divmod(int(n_items), int(n_slots))
    begin = slot * base + min(slot, rem)
    return begin, begin + base + (1 if slot < rem else 0)


class Candidates(object):
    """robo_cand: a device-resident candidate batch (normalised input space) + workspace."""

    def __init__(self, ctx, Xc=None, m=None, dim=None, seed=None, n_uniform=None, loc=None, scale=None, sobol=None,
                 first=0):
        self.ctx = ctx
        self._h = C.c_void_p()
        self._owner = lib()
        if sobol is not None:
            # sobol: a scipy.stats.qmc.Sobol engine (its direction numbers and digital shift are read, the engine is
            # not advanced) or a (sv (dim, bits), shift (dim,), bits) triple; points first .. first + m - 1
            if hasattr(sobol, "random"):          # a scipy.stats.qmc.Sobol engine
                if not all(hasattr(sobol, a) for a in ("_sv", "_shift", "bits")):
                    raise RoboHipError("this SciPy's qmc.Sobol does not expose its direction numbers (_sv, _shift, "
                                       "bits; SciPy 1.7-1.15 do): pass a (sv, shift, bits) triple instead")
                sv, shift, bits = sobol._sv, sobol._shift, int(sobol.bits)
            else:
                sv, shift, bits = sobol
            sv = np.ascontiguousarray(sv, dtype=np.uint64)
            shift = np.ascontiguousarray(shift, dtype=np.uint64)
            assert sv.ndim == 2 and sv.shape[1] == bits and shift.shape == (sv.shape[0],)
            self.m, self.dim = int(m), int(sv.shape[0])
            u64p = C.POINTER(C.c_uint64)
            check(lib().robo_cand_create_sobol(ctx._h, self.m, self.dim, sv.ctypes.data_as(u64p),
                                               shift.ctypes.data_as(u64p), bits, int(first), C.byref(self._h)))
        elif loc is not None:
            loc, scale = _f64(loc), _f64(scale)
            self.m, self.dim = int(m), int(loc.shape[0])
            assert scale.shape == loc.shape
            check(lib().robo_cand_create_random(ctx._h, self.m, self.dim, int(seed or 0),
                                                self.m if n_uniform is None else int(n_uniform), _arr(loc),
                                                _arr(scale), C.byref(self._h)))
        elif Xc is not None:
            Xc = _f64(Xc)
            assert Xc.ndim == 2
            self.m, self.dim = Xc.shape
            check(lib().robo_cand_create(ctx._h, _arr(Xc), self.m, self.dim, C.byref(self._h)))
        else:
            self.m, self.dim = int(m), int(dim)
            check(lib().robo_cand_create_uniform(ctx._h, self.m, self.dim, int(seed or 0), C.byref(self._h)))

    def set_points(self, Xc):
        """upload a new batch of the same shape into this handle (H2D only)"""
        Xc = _f64(Xc, (self.m, self.dim))
        check(lib().robo_cand_set_points(self._h, _arr(Xc), self.m))

    def point(self, index):
        out = np.empty(self.dim)
        check(lib().robo_cand_get_point(self._h, int(index), _arr(out)))
        return out

    def chunk(self):
        """candidates per workspace pass of the last posterior evaluated on this handle (0: none yet)"""
        n = C.c_int64(0)
        check(lib().robo_cand_workspace_chunk(self._h, C.byref(n)))
        return int(n.value)

    def solve_kernel(self):
        """name of the kernel that ran the solve of the last posterior evaluated on this handle"""
        buf = C.create_string_buffer(64)
        check(lib().robo_cand_last_solve_kernel(self._h, buf, 64))
        return buf.value.decode()

    def points(self):
        out = np.empty((self.m, self.dim))
        check(lib().robo_cand_get_points(self._h, _arr(out)))
        return out

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._owner.robo_cand_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _full_theta(gp, theta):
    """the hyper-parameter vector(s) the library takes, from what the caller holds: ``gp.fixed_head`` in front.  A george
    kernel WITHOUT an amplitude factor has no amplitude parameter (``Matern52Kernel(metric, ndim)`` alone: len = D; with
    ``amp * kernel`` in front: 1 + D) -- the library's vector always starts with log amp, which such a caller pins at 0."""
    head = gp.fixed_head
    if not head:
        return theta
    theta = np.asarray(theta, dtype=np.float64)
    if theta.ndim == 1:
        return np.concatenate([head, theta])
    return np.hstack([np.tile(np.asarray(head, dtype=np.float64), (theta.shape[0], 1)), theta])


class DeviceGP(object):
    """robo_gp: training data + Cholesky factor + z resident on the device."""

    def __init__(self, ctx, kind, n_max, dim, fixed_head=()):
        self.ctx, self.kind, self.n_max, self.dim = ctx, kind, int(n_max), int(dim)
        self._h = C.c_void_p()
        self._owner = lib()
        check(self._owner.robo_gp_create(ctx._h, KERNEL_KINDS[kind], self.n_max, self.dim, C.byref(self._h)))
        self.n = 0
        self.n_theta = lib().robo_theta_size(KERNEL_KINDS[kind], self.dim)
        # leading library hyper-parameters the caller does not hold (see _full_theta); () for every kernel with an amplitude
        self.fixed_head = tuple(float(v) for v in fixed_head)

    def set_precision(self, fp32_gram):
        check(lib().robo_gp_set_precision(self._h, 1 if fp32_gram else 0))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._owner.robo_gp_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_data(self, X, y):
        X, y = _f64(X), _f64(y)
        assert X.ndim == 2 and y.ndim == 1 and X.shape[0] == y.shape[0] and X.shape[1] == self.dim
        check(lib().robo_gp_set_data(self._h, _arr(X), _arr(y), X.shape[0]))
        self.n = X.shape[0]

    def set_output_transform(self, y_mean, y_std):
        check(lib().robo_gp_set_output_transform(self._h, float(y_mean), float(y_std)))

    def fit(self, theta, mean_c):
        """-> log-likelihood; raises np.linalg.LinAlgError when K is not PD."""
        theta = _f64(_full_theta(self, theta), (self.n_theta,))
        ll, col = C.c_double(0), C.c_int32(0)
        check(lib().robo_gp_fit(self._h, _arr(theta), float(mean_c), C.byref(ll), C.byref(col)))
        return ll.value

    def grad_loglik(self, theta, mean_c):
        """(log likelihood, d loglik / d theta) at theta; the GP is fitted at theta afterwards.  The last
        gradient entry is d / d sigma^2, as in the reference's grad_nll (include/robo_hip.h)."""
        theta = _f64(_full_theta(self, theta), (self.n_theta,))
        ll = C.c_double()
        col = C.c_int32()
        grad = np.empty(self.n_theta)
        check(lib().robo_gp_grad_loglik(self._h, _arr(theta), float(mean_c), C.byref(ll), _arr(grad), C.byref(col)))
        return ll.value, grad[len(self.fixed_head):]

    def loglik_batch(self, thetas, mean_c):
        thetas = _f64(_full_theta(self, np.atleast_2d(thetas)))
        assert thetas.ndim == 2 and thetas.shape[1] == self.n_theta
        S = thetas.shape[0]
        ll = np.empty(S)
        st = np.empty(S, dtype=np.int32)
        check(lib().robo_gp_loglik_batch(self._h, _arr(thetas), S, float(mean_c), _arr(ll),
                                         st.ctypes.data_as(C.POINTER(C.c_int32))))
        return ll, st

    def mcmc_run(self, mean_c, prior, pos, lnp, n_steps, u_stretch, partner, u_accept, a=2.0):
        """emcee 2's EnsembleSampler.run_mcmc on the device (robo_gp_mcmc_run): prior = None, (1, 5 parameters) DefaultPrior or
        (2, 9 parameters) EnvPrior;
        lnp None = evaluate the start positions.  -> (pos, lnp, chain (k, n_steps, P), lnprob (k, n_steps), accepted (k))"""
        pos = np.array(pos, dtype=np.float64, order="C")
        k = pos.shape[0]
        assert not self.fixed_head, "the device chain moves every library hyper-parameter: use the host sampler"
        assert pos.shape == (k, self.n_theta)
        eval_start = lnp is None
        lnp = np.zeros(k) if eval_start else np.array(lnp, dtype=np.float64)
        n_steps = int(n_steps)
        uz = np.ascontiguousarray(u_stretch, dtype=np.float64).reshape(-1)
        ua = np.ascontiguousarray(u_accept, dtype=np.float64).reshape(-1)
        pa = np.ascontiguousarray(partner, dtype=np.int32).reshape(-1)
        assert uz.size == ua.size == pa.size == n_steps * k
        chain = np.empty((k, n_steps, self.n_theta))
        lnps = np.empty((k, n_steps))
        acc = np.zeros(k, dtype=np.int64)
        if prior is None:
            kind, par = 0, np.zeros(9)
        else:                                    # (kind, 5 parameters) DefaultPrior / (kind, 9 parameters) EnvPrior
            kind = int(prior[0])
            par = np.zeros(9)
            given = _f64(prior[1]).reshape(-1)
            assert given.size == (9 if kind == 2 else 5)
            par[:given.size] = given
        check(lib().robo_gp_mcmc_run(self._h, float(mean_c), kind, _arr(par), k, n_steps, float(a), _arr(uz),
                                     pa.ctypes.data_as(C.POINTER(C.c_int32)), _arr(ua), int(eval_start), _arr(pos),
                                     _arr(lnp), _arr(chain), _arr(lnps), acc.ctypes.data_as(C.POINTER(C.c_int64))))
        return pos, lnp, chain, lnps, acc

    def factor(self):
        out = np.empty((self.n, self.n))
        check(lib().robo_gp_get_factor(self._h, _arr(out)))
        return out

    def prefetch_inverse(self):
        """start building W = L^-1 for small candidate batches now, asynchronously (robo_gp_prefetch_inverse)"""
        check(lib().robo_gp_prefetch_inverse(self._h))

    def factor_cond(self):
        """(cond_inf(L) -- exact, from the explicit inverse --, min L_ii, max L_ii) of the current factor"""
        out = np.zeros(3)
        check(lib().robo_gp_factor_cond(self._h, _arr(out)))
        return float(out[0]), float(out[1]), float(out[2])

    def gram(self, theta):
        theta = _f64(_full_theta(self, theta), (self.n_theta,))
        out = np.empty((self.n, self.n))
        check(lib().robo_gp_get_gram(self._h, _arr(theta), _arr(out)))
        return out

    def diag_timeline(self, theta):
        theta = _f64(theta, (self.n_theta,))
        out = np.zeros(38)
        check(diag().robo_selftest_diag_timeline(self._h, _arr(theta), _arr(out)))
        return out

    def predict(self, Xc):
        if isinstance(Xc, Candidates):
            mean, var = np.empty(Xc.m), np.empty(Xc.m)
            check(lib().robo_gp_predict_cand(self._h, Xc._h, _arr(mean), _arr(var)))
            return mean, var
        Xc = _f64(Xc)
        assert Xc.ndim == 2 and Xc.shape[1] == self.dim
        mean, var = np.empty(Xc.shape[0]), np.empty(Xc.shape[0])
        check(lib().robo_gp_predict(self._h, _arr(Xc), Xc.shape[0], _arr(mean), _arr(var)))
        return mean, var

    def predict_cov(self, Xc):
        Xc = _f64(Xc)
        assert Xc.ndim == 2 and Xc.shape[1] == self.dim
        m = Xc.shape[0]
        mean, cov = np.empty(m), np.empty((m, m))
        check(lib().robo_gp_predict_cov(self._h, _arr(Xc), m, _arr(mean), _arr(cov)))
        return mean, cov

    def predict_grad(self, Xc):
        """-> (mean (M,), var (M,), d mean / d x (M, D), d var / d x (M, D)) in the GP's input space"""
        Xc = _f64(Xc)
        assert Xc.ndim == 2 and Xc.shape[1] == self.dim
        m = Xc.shape[0]
        mean, var, dm, dv = np.empty(m), np.empty(m), np.empty((m, self.dim)), np.empty((m, self.dim))
        check(lib().robo_gp_predict_grad(self._h, _arr(Xc), m, _arr(mean), _arr(var), _arr(dm), _arr(dv)))
        return mean, var, dm, dv

    def acq(self, kind, par, eta, Xc, want_values=True):
        """-> (values or None, max, argmax, flags)"""
        mx, am, fl = C.c_double(0), C.c_int64(0), C.c_uint32(0)
        if isinstance(Xc, Candidates):
            m = Xc.m
            out = np.empty(m) if want_values else None
            check(lib().robo_acq_eval_cand(self._h, ACQ_KINDS[kind], float(par), float(eta), Xc._h,
                                           _arr(out) if want_values else None, C.byref(mx), C.byref(am),
                                           C.byref(fl)))
        else:
            Xc = _f64(Xc)
            assert Xc.ndim == 2 and Xc.shape[1] == self.dim
            m = Xc.shape[0]
            out = np.empty(m) if want_values else None
            check(lib().robo_acq_eval(self._h, ACQ_KINDS[kind], float(par), float(eta), _arr(Xc), m,
                                      _arr(out) if want_values else None, C.byref(mx), C.byref(am), C.byref(fl)))
        return out, mx.value, am.value, fl.value


class Comm(object):
    """robo_comm: this process's rank in a one-process-per-GPU job; the exchanges of the candidate / sample shards run
    inside the library as RCCL all-gathers on the context's stream (include/robo_hip.h).  Every method is COLLECTIVE."""

    @staticmethod
    def create_id():
        """rank 0: the 128-byte id every rank must be given (out of band) before constructing its Comm"""
        buf = C.create_string_buffer(COMM_ID_BYTES)
        check(lib().robo_comm_create_id(buf))
        return bytes(buf.raw)

    def __init__(self, ctx, rank, world, comm_id):
        assert len(comm_id) == COMM_ID_BYTES
        self.ctx, self.rank, self.world = ctx, int(rank), int(world)
        self._h = C.c_void_p()
        self._owner = lib()
        check(self._owner.robo_comm_init(ctx._h, self.rank, self.world, bytes(comm_id), C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._owner.robo_comm_destroy(self._h)
            self._h = None

    def info(self):
        """(rank, world) as the communicator holds them (robo_comm_info)"""
        r, w = C.c_int32(-1), C.c_int32(-1)
        check(lib().robo_comm_info(self._h, C.byref(r), C.byref(w)))
        return int(r.value), int(w.value)

    def allgather(self, values):
        """values: (count,) host doubles -> (world, count), identical on every rank"""
        v = _f64(np.asarray(values, dtype=np.float64).reshape(-1))
        out = np.empty((self.world, v.shape[0]))
        check(lib().robo_comm_allgather(self._h, _arr(v), v.shape[0], _arr(out)))
        return out

    def acq_sharded(self, gp, kind, par, eta, cand, global_offset, want_values=False):
        """candidate shard -> (this rank's values or None, GLOBAL max, GLOBAL argmax, owner rank, flags OR-ed)"""
        out = np.empty(cand.m) if want_values else None
        mx, am, own, fl = C.c_double(0), C.c_int64(0), C.c_int32(0), C.c_uint32(0)
        check(lib().robo_acq_eval_cand_sharded(self._h, gp._h, ACQ_KINDS[kind], float(par), float(eta), cand._h,
                                               int(global_offset), _arr(out) if want_values else None, C.byref(mx),
                                               C.byref(am), C.byref(own), C.byref(fl)))
        return out, mx.value, am.value, own.value, fl.value

    def ig_per_cost_sharded(self, gp, cand, rep, ep, sn2, cost_gp, cost_cand, overhead, global_offset,
                            want_values=False):
        """candidate shard of the information gain per unit cost -> (this rank's values or None, GLOBAL max, GLOBAL
        argmax, owner rank)"""
        assert rep.m == ep.nb and cost_cand.m == cand.m
        out = np.empty(cand.m) if want_values else None
        mx, am, own = C.c_double(0), C.c_int64(0), C.c_int32(0)
        check(lib().robo_ig_eval_per_cost_cand_sharded(self._h, gp._h, cand._h, rep._h, ep.W.size, float(sn2),
                                                       *ep.args(), cost_gp._h, cost_cand._h, float(overhead),
                                                       int(global_offset), _arr(out) if want_values else None,
                                                       C.byref(mx), C.byref(am), C.byref(own)))
        return out, mx.value, am.value, own.value

    def acq_marginal_sharded(self, gps, s_total, kind, par, etas, cand, want_values=True):
        """sample shard: this rank's fitted GPs (possibly none) -> (mean over ALL s_total samples or None, max,
        argmax, flags), identical on every rank"""
        S = len(gps)
        arr = (C.c_void_p * max(S, 1))(*[g._h for g in gps])
        etas = _f64(np.asarray(etas, dtype=np.float64).reshape(-1)) if S else np.zeros(1)
        out = np.empty(cand.m) if want_values else None
        mx, am, fl = C.c_double(0), C.c_int64(0), C.c_uint32(0)
        check(lib().robo_acq_eval_marginal_cand_sharded(self._h, arr, S, int(s_total), ACQ_KINDS[kind], float(par),
                                                        _arr(etas), cand._h, _arr(out) if want_values else None,
                                                        C.byref(mx), C.byref(am), C.byref(fl)))
        return out, mx.value, am.value, fl.value


class CandidateShards(object):
    """the candidate batch of ONE maximisation split over the contexts of a Multi: shards[g] is a Candidates on context g
    (or None: empty shard), offsets[g] the global index of its first row"""

    def __init__(self, shards, offsets):
        self.shards, self.offsets = list(shards), [int(o) for o in offsets]
        self.m = sum(c.m for c in self.shards if c is not None)

    @classmethod
    def split(cls, ctxs, Xn):
        """contiguous shards of the host matrix Xn (already in the model's input space), one upload per device"""
        Xn = _f64(Xn)
        shards, offsets = [], []
        for g, ctx in enumerate(ctxs):
            b, e = shard_range(Xn.shape[0], g, len(ctxs))
            offsets.append(b)
            shards.append(Candidates(ctx, Xn[b:e]) if e > b else None)
        return cls(shards, offsets)

    def owner_point(self, owner, global_index):
        return self.shards[owner].point(int(global_index) - self.offsets[owner])

    def point(self, global_index):
        """row `global_index` of the whole batch, from whichever shard holds it"""
        for g, c in enumerate(self.shards):
            if c is not None and self.offsets[g] <= global_index < self.offsets[g] + c.m:
                return self.owner_point(g, global_index)
        raise IndexError(global_index)

    def close(self):
        for c in self.shards:
            if c is not None:
                c.close()


class Multi(object):
    """robo_multi: G contexts of THIS process, one per device; every method fans its shards out to all devices at once
    (one worker thread per device inside the library) and returns the reduced result (include/robo_hip.h)."""

    def __init__(self, ctxs):
        self.ctxs = list(ctxs)
        self.n = len(self.ctxs)
        arr = (C.c_void_p * self.n)(*[c._h for c in self.ctxs])
        self._h = C.c_void_p()
        self._lib = lib()            # the library that owns the handle (tests switch libraries: use_library)
        check(self._lib.robo_multi_create(arr, self.n, C.byref(self._h)))
        _live_multis.add(self)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.robo_multi_destroy(self._h)      # joins the worker threads
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def info(self):
        """(number of contexts, their HIP devices, worker threads in use)"""
        n, t = C.c_int32(0), C.c_int32(0)
        dev = (C.c_int32 * self.n)()
        check(lib().robo_multi_info(self._h, C.byref(n), dev, C.byref(t)))
        return int(n.value), [int(d) for d in dev], int(t.value)

    def _handles(self, objs):
        return (C.c_void_p * max(len(objs), 1))(*[(o._h if o is not None else None) for o in objs])

    @staticmethod
    def _flatten(groups):
        flat = [g for grp in groups for g in grp]
        counts = (C.c_int32 * len(groups))(*[len(grp) for grp in groups])
        return flat, counts

    def set_data(self, gps, X, y):
        X, y = _f64(X), _f64(y)
        assert len(gps) == self.n and X.ndim == 2 and y.shape == (X.shape[0],)
        check(lib().robo_gp_set_data_multi(self._h, self._handles(gps), _arr(X), _arr(y), X.shape[0]))
        for g in gps:
            g.n = X.shape[0]

    def fit(self, gps, theta, mean_c):
        """the same fit on every device's replica -> log-likelihood; raises np.linalg.LinAlgError like DeviceGP.fit"""
        assert len(gps) == self.n
        theta = _f64(_full_theta(gps[0], theta), (gps[0].n_theta,))
        ll, col = C.c_double(0), C.c_int32(0)
        check(lib().robo_gp_fit_multi(self._h, self._handles(gps), _arr(theta), float(mean_c), C.byref(ll), C.byref(col)))
        return ll.value

    def loglik_batch(self, gps, thetas, mean_c):
        """DeviceGP.loglik_batch with the thetas split over the devices (gps: one data-holding handle per device)"""
        assert len(gps) == self.n
        thetas = _f64(_full_theta(gps[0], np.atleast_2d(thetas)))
        S = thetas.shape[0]
        ll = np.empty(S)
        st = np.empty(S, dtype=np.int32)
        check(lib().robo_gp_loglik_batch_multi(self._h, self._handles(gps), _arr(thetas), S, float(mean_c), _arr(ll),
                                               st.ctypes.data_as(C.POINTER(C.c_int32))))
        return ll, st

    def fit_batch(self, gp_groups, thetas, mean_c):
        """fit_batch per device: gp_groups[g] = the handles of device g (its first one holds the data), thetas in the
        flattened order -> (loglik, status)"""
        assert len(gp_groups) == self.n
        flat, counts = self._flatten(gp_groups)
        thetas = _f64(_full_theta(flat[0], np.atleast_2d(thetas))) if flat else _f64(thetas)
        assert thetas.shape[0] == len(flat)
        ll = np.empty(len(flat))
        st = np.empty(len(flat), dtype=np.int32)
        check(lib().robo_gp_fit_batch_multi(self._h, self._handles(flat), counts, _arr(thetas), float(mean_c), _arr(ll),
                                            st.ctypes.data_as(C.POINTER(C.c_int32))))
        n = next((grp[0].n for grp in gp_groups if grp), 0)
        for g in flat:
            g.n = n
        return ll, st

    def acq(self, gps, kind, par, eta, shards, want_values=False):
        """candidate shard -> (values of all shards in slot order or None, max, GLOBAL argmax, owner slot, flags OR-ed)"""
        assert len(gps) == self.n and len(shards.shards) == self.n
        out = np.empty(shards.m) if want_values else None
        mx, am, own, fl = C.c_double(0), C.c_int64(0), C.c_int32(0), C.c_uint32(0)
        offs = (C.c_int64 * self.n)(*shards.offsets)
        check(lib().robo_acq_eval_cand_multi(self._h, self._handles(gps), ACQ_KINDS[kind], float(par), float(eta),
                                             self._handles(shards.shards), offs, _arr(out) if want_values else None,
                                             C.byref(mx), C.byref(am), C.byref(own), C.byref(fl)))
        return out, mx.value, am.value, own.value, fl.value

    def ig(self, gps, shards, reps, ep, sn2, want_values=True):
        """candidate shard of the information gain -> (values of all shards in slot order or None, max, GLOBAL argmax, owner)"""
        assert all(len(x) == self.n for x in (gps, shards.shards, reps))
        out = np.empty(shards.m) if want_values else None
        mx, am, own = C.c_double(0), C.c_int64(0), C.c_int32(0)
        offs = (C.c_int64 * self.n)(*shards.offsets)
        check(lib().robo_ig_eval_cand_multi(self._h, self._handles(gps), self._handles(shards.shards), self._handles(reps),
                                            ep.W.size, float(sn2), *ep.args(), offs, _arr(out) if want_values else None,
                                            C.byref(mx), C.byref(am), C.byref(own)))
        return out, mx.value, am.value, own.value

    def ig_per_cost(self, gps, shards, reps, ep, sn2, cost_gps, cost_shards, overhead, want_values=False):
        """candidate shard of the information gain per unit cost -> (values or None, max, GLOBAL argmax, owner slot)"""
        assert all(len(x) == self.n for x in (gps, shards.shards, reps, cost_gps, cost_shards.shards))
        out = np.empty(shards.m) if want_values else None
        mx, am, own = C.c_double(0), C.c_int64(0), C.c_int32(0)
        offs = (C.c_int64 * self.n)(*shards.offsets)
        check(lib().robo_ig_eval_per_cost_cand_multi(self._h, self._handles(gps), self._handles(shards.shards),
                                                     self._handles(reps), ep.W.size, float(sn2), *ep.args(),
                                                     self._handles(cost_gps), self._handles(cost_shards.shards),
                                                     float(overhead), offs, _arr(out) if want_values else None,
                                                     C.byref(mx), C.byref(am), C.byref(own)))
        return out, mx.value, am.value, own.value

    def acq_marginal(self, gp_groups, kind, par, eta_groups, cands, want_values=True):
        """sample shard: gp_groups[g] / eta_groups[g] = the fitted handles of device g and their incumbent values,
        cands[g] = ALL candidates on device g (None where a device has no sample, g > 0) -> (mean over all samples or
        None, max, argmax, flags)"""
        assert len(gp_groups) == self.n and len(cands) == self.n
        flat, counts = self._flatten(gp_groups)
        etas = _f64(np.concatenate([np.asarray(e, dtype=np.float64).reshape(-1) for e in eta_groups] or [np.zeros(0)]))
        assert etas.shape[0] == len(flat)
        out = np.empty(cands[0].m) if want_values else None
        mx, am, fl = C.c_double(0), C.c_int64(0), C.c_uint32(0)
        check(lib().robo_acq_eval_marginal_cand_multi(self._h, self._handles(flat), counts, ACQ_KINDS[kind], float(par),
                                                      _arr(etas), self._handles(cands),
                                                      _arr(out) if want_values else None, C.byref(mx), C.byref(am),
                                                      C.byref(fl)))
        return out, mx.value, am.value, fl.value

    def predict_mixture(self, gp_groups, cands):
        """GaussianProcessMCMC.predict over samples that live on several devices -> (mean (M,), var (M,))"""
        assert len(gp_groups) == self.n and len(cands) == self.n
        flat, counts = self._flatten(gp_groups)
        mean, var = np.empty(cands[0].m), np.empty(cands[0].m)
        check(lib().robo_gp_predict_mixture_cand_multi(self._h, self._handles(flat), counts, self._handles(cands),
                                                       _arr(mean), _arr(var)))
        return mean, var


def fit_batch(gps, thetas, mean_c):
    """GaussianProcessMCMC.train's per-sample fits in one batched pass that keeps the factors:
    gps[0] holds the data; afterwards gps[s] is fitted at thetas[s] wherever status[s] == OK.
    -> (loglik (S,), status (S,))"""
    S = len(gps)
    thetas = _f64(_full_theta(gps[0], np.atleast_2d(thetas)))
    assert thetas.shape == (S, gps[0].n_theta)
    arr = (C.c_void_p * S)(*[g._h for g in gps])
    ll = np.empty(S)
    st = np.empty(S, dtype=np.int32)
    check(lib().robo_gp_fit_batch(arr, S, _arr(thetas), float(mean_c), _arr(ll), st.ctypes.data_as(C.POINTER(C.c_int32))))
    for g in gps:
        g.n = gps[0].n
    return ll, st


def predict_mixture(gps, cand):
    """GaussianProcessMCMC.predict over device GPs -> (mean (M,), var (M,))"""
    S = len(gps)
    arr = (C.c_void_p * S)(*[g._h for g in gps])
    mean, var = np.empty(cand.m), np.empty(cand.m)
    check(lib().robo_gp_predict_mixture_cand(arr, S, cand._h, _arr(mean), _arr(var)))
    return mean, var


def acq_from_moments(ctx, kind, par, eta, mean, var):
    """element-wise acquisition on the device for (mean, var) of any model -> (values, max, argmax, flags)"""
    mean, var = _f64(mean), _f64(var)
    assert mean.ndim == 1 and mean.shape == var.shape
    out = np.empty(mean.shape[0])
    mx, am, fl = C.c_double(0), C.c_int64(0), C.c_uint32(0)
    check(lib().robo_acq_eval_moments(ctx._h, ACQ_KINDS[kind], float(par), float(eta), _arr(mean), _arr(var),
                                      mean.shape[0], _arr(out), C.byref(mx), C.byref(am), C.byref(fl)))
    return out, mx.value, am.value, fl.value


class EPState(object):
    """the host-side EP tensors of entropy search, contiguous fp64"""

    def __init__(self, logP, lmb, W, dlogPdMu, dlogPdSigma, dlogPdMudMu):
        self.nb = int(np.asarray(logP).size)
        self.logP = _f64(np.asarray(logP).reshape(-1))
        self.lmb = _f64(np.asarray(lmb).reshape(-1))
        self.W = _f64(np.asarray(W).reshape(-1))
        self.dMu = _f64(dlogPdMu, (self.nb, self.nb))
        self.dSigma = _f64(dlogPdSigma, (self.nb, self.nb * (self.nb + 1) // 2))
        self.dMuMu = _f64(dlogPdMudMu, (self.nb, self.nb, self.nb))

    def args(self):
        return [_arr(a) for a in (self.logP, self.lmb, self.W, self.dMu, self.dSigma, self.dMuMu)]


def ig_eval(gp, cand, rep, ep, sn2, want_values=True):
    """information gain of every candidate -> (values, max, argmax)"""
    assert rep.m == ep.nb
    out = np.empty(cand.m) if want_values else None
    mx, am = C.c_double(0), C.c_int64(0)
    check(lib().robo_ig_eval_cand(gp._h, cand._h, rep._h, ep.W.size, float(sn2), *ep.args(),
                                  _arr(out) if want_values else None, C.byref(mx), C.byref(am)))
    return out, mx.value, am.value


def ig_eval_per_cost(gp, cand, rep, ep, sn2, cost_gp, cost_cand, overhead=0.0, want_values=True):
    """information gain per unit cost of every candidate, dH / (exp(cost mean) + overhead), and its argmax, in one
    library call -> (values or None, max, argmax)"""
    assert rep.m == ep.nb and cost_cand.m == cand.m
    out = np.empty(cand.m) if want_values else None
    mx, am = C.c_double(0), C.c_int64(0)
    check(lib().robo_ig_eval_per_cost_cand(gp._h, cand._h, rep._h, ep.W.size, float(sn2), *ep.args(), cost_gp._h,
                                           cost_cand._h, float(overhead), _arr(out) if want_values else None,
                                           C.byref(mx), C.byref(am)))
    return out, mx.value, am.value


def ig_from_moments(ctx, s, v, ep, sn2):
    s, v = _f64(s), _f64(v)
    assert s.ndim == 2 and s.shape[1] == ep.nb and v.shape == (s.shape[0],)
    out = np.empty(s.shape[0])
    check(lib().robo_ig_eval_moments(ctx._h, s.shape[0], ep.nb, ep.W.size, float(sn2), _arr(s), _arr(v), *ep.args(),
                                     _arr(out)))
    return out


def cross_cov(gp, cand, ref):
    out = np.empty((cand.m, ref.m))
    check(lib().robo_gp_cross_cov(gp._h, cand._h, ref._h, _arr(out)))
    return out


def acq_marginal(gps, kind, par, eta, cand, want_values=True, reduce="mean"):
    """MarginalizationGPMCMC.compute over device GPs -> (values, max, argmax, flags).
    eta: one incumbent value for all samples, or one per sample (S,)."""
    S = len(gps)
    arr = (C.c_void_p * S)(*[g._h for g in gps])
    etas = _f64(np.broadcast_to(np.asarray(eta, dtype=np.float64), (S,)))
    mx, am, fl = C.c_double(0), C.c_int64(0), C.c_uint32(0)
    out = np.empty(cand.m) if (want_values or reduce == "sum") else None
    if reduce == "sum":
        check(lib().robo_acq_eval_sum_cand(arr, S, ACQ_KINDS[kind], float(par), _arr(etas), cand._h, _arr(out),
                                           C.byref(fl)))
        return out, None, None, fl.value
    check(lib().robo_acq_eval_marginal_cand(arr, S, ACQ_KINDS[kind], float(par), _arr(etas), cand._h,
                                            _arr(out) if want_values else None, C.byref(mx), C.byref(am),
                                            C.byref(fl)))
    return out, mx.value, am.value, fl.value

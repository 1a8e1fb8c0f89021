"""Build the TEST-ONLY g++ interpretation of the HIP sources (tests/hipemu/README.md).

The product sources under robo_amd/csrc are compiled unmodified against the stand-in
<hip/hip_runtime.h> of this directory.  Output: tests/hipemu/_build/librobo_emu.so.
Never loaded by robo_amd; only by tests that pass an explicit library path.
"""
import contextlib
import fcntl
import glob
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
OUT = os.path.join(HERE, "_build", "librobo_emu.so")
# The CONTRACTING variant: the device compiler's own front end (clang) on the host with its default contraction rule
# (-ffp-contract=fast-honor-pragmas) and x86 FMA enabled, so every  a * b + c  the device compiler may fuse IS fused
# here too, and `#pragma clang fp contract(off)` means what it means on the device.  g++ (the variant above) never
# fuses and ignores that pragma: it is stricter than the MI355X and hid round 5's fused stretch-move proposal.
OUT_FMA = os.path.join(HERE, "_build", "librobo_emu_fma.so")
HOST_CLANG = os.environ.get("HIPEMU_CLANG", "/opt/rocm/lib/llvm/bin/clang++")


@contextlib.contextmanager
def _build_lock():
    """one builder at a time (several test processes may find the library stale at the same moment)"""
    os.makedirs(os.path.join(HERE, "_build"), exist_ok=True)
    with open(os.path.join(HERE, "_build", ".lock"), "w") as f:
        fcntl.flock(f, fcntl.LOCK_EX)
        try:
            yield
        finally:
            fcntl.flock(f, fcntl.LOCK_UN)


def build(force=False):
    with _build_lock():
        return _build(force)


def build_fma(force=False):
    """the contracting interpreter (see OUT_FMA)"""
    with _build_lock():
        return _build(force, out=OUT_FMA, compiler=[HOST_CLANG, "-ffp-contract=fast-honor-pragmas", "-mfma",
                                                    "-Wno-pass-failed"], tag="fma.")


def _build(force=False, out=OUT, compiler=("g++", "-Wno-psabi"), tag=""):
    OUT = out
    srcs = sorted(glob.glob(os.path.join(ROOT, "robo_amd", "csrc", "*.hip")) +
                  glob.glob(os.path.join(ROOT, "robo_amd", "csrc", "diag", "*.hip")))   # one interpreter library
    deps = srcs + glob.glob(os.path.join(ROOT, "robo_amd", "csrc", "*.h")) + \
        [os.path.join(HERE, "hipemu.cpp"), os.path.join(HERE, "hip", "hip_runtime.h"),
         os.path.join(ROOT, "include", "robo_hip.h"), os.path.join(ROOT, "include", "robo_hip_diag.h")]
    if not force and os.path.exists(OUT) and all(os.path.getmtime(d) <= os.path.getmtime(OUT) for d in deps):
        return OUT
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    objs = []
    for s in srcs + [os.path.join(HERE, "hipemu.cpp")]:
        o = os.path.join(HERE, "_build", tag + os.path.basename(s) + ".o")
        cmd = list(compiler) + ["-O2", "-g", "-std=c++17", "-fPIC", "-Wno-unknown-pragmas", "-I", HERE,
                                "-x", "c++", "-c", s, "-o", o]
        subprocess.check_call(cmd)
        objs.append(o)
    subprocess.check_call([compiler[0], "-shared", "-o", OUT] + objs)
    return OUT


FAKE_RCCL = os.path.join(HERE, "_build", "libfake_rccl.so")


def build_fake_rccl(force=False):
    """the shared-memory stand-in for librccl.so (fake_rccl.cpp): ROBO_RCCL_LIB points comm.hip at it"""
    with _build_lock():
        return _build_fake_rccl(force)


def _build_fake_rccl(force=False):
    src = os.path.join(HERE, "fake_rccl.cpp")
    if not force and os.path.exists(FAKE_RCCL) and os.path.getmtime(src) <= os.path.getmtime(FAKE_RCCL):
        return FAKE_RCCL
    os.makedirs(os.path.dirname(FAKE_RCCL), exist_ok=True)
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", src, "-o", FAKE_RCCL, "-lrt", "-pthread"])
    return FAKE_RCCL


SCHED_SELFTEST = os.path.join(HERE, "_build", "libsched_selftest.so")


def build_sched_selftest(force=False):
    """sched_selftest.cpp + its own copy of the interpreter: the deferred stream schedules checked on a known race"""
    with _build_lock():
        return _build_sched_selftest(force)


def _build_sched_selftest(force=False):
    srcs = [os.path.join(HERE, "sched_selftest.cpp"), os.path.join(HERE, "hipemu.cpp")]
    deps = srcs + [os.path.join(HERE, "hip", "hip_runtime.h")]
    if not force and os.path.exists(SCHED_SELFTEST) and all(os.path.getmtime(d) <= os.path.getmtime(SCHED_SELFTEST) for d in deps):
        return SCHED_SELFTEST
    os.makedirs(os.path.dirname(SCHED_SELFTEST), exist_ok=True)
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fPIC", "-shared", "-Wno-psabi", "-I", HERE] + srcs +
                          ["-o", SCHED_SELFTEST])
    return SCHED_SELFTEST


if __name__ == "__main__":
    print(build(force=True))

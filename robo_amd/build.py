"""Build librobo_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python -m robo_amd.build [--force]

The library is built IN TREE (robo_amd/librobo_hip.so) so that it travels to the GPU box
with the repository snapshot.  There is no CPU build of this library: robo_amd fails
loudly when it is missing.
"""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "librobo_hip.so")
# self-checks / micro-benchmarks (include/robo_hip_diag.h): test and measurement infrastructure, a separate library
DIAG_LIB = os.path.join(HERE, "librobo_hip_diag.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-function", "-Wno-unused-value",
         "-munsafe-fp-atomics", "-mllvm", "-amdgpu-mfma-vgpr-form"]


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def diag_sources():
    return sorted(glob.glob(os.path.join(CSRC, "diag", "*.hip")))


def _stale():
    if not os.path.exists(LIB) or not os.path.exists(DIAG_LIB):
        return True
    t = min(os.path.getmtime(LIB), os.path.getmtime(DIAG_LIB))
    deps = sources() + diag_sources() + glob.glob(os.path.join(CSRC, "*.h")) + \
        glob.glob(os.path.join(HERE, "..", "include", "*.h"))
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True, extra_flags=()):
    """-> path of the product library; the diagnostics library is built next to it"""
    if not force and not _stale():
        return LIB
    for out, srcs, link in ((LIB, sources(), []),
                            (DIAG_LIB, diag_sources(), ["-L" + HERE, "-l:librobo_hip.so", "-Wl,-rpath,$ORIGIN"])):
        cmd = [HIPCC] + FLAGS + list(extra_flags) + srcs + link + ["-o", out]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)

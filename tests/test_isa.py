"""Compiler-output checks (CPU: hipcc cross-compiles gfx950 without a GPU).

Round 5's hardware failure: `c - z (c - s)` of the stretch move was written with ROCm's __dmul_rn / __dsub_rn, which are
the plain operators, and hipcc's default -ffp-contract=fast-honor-pragmas compiled it to  v_fma_f64 q = -z t + c.  NumPy
(emcee) rounds z (c - s) first; the one-ulp seeds grew along the chain and the walkers left the reference's after a few
hundred steps on the MI355X only (the g++ interpreter never fuses).  These tests read what the compiler emits:

* LLVM IR of every kernel that inlines the proposal (mcmc.hip, potrf.hip): the three operations of q, the five of z and
  the three of the accept statistic carry no `contract` flag -- the only thing the AMDGPU backend may fuse under
  fast-honor-pragmas -- and no fma / fmuladd intrinsic sits on those paths.
* machine code of the probe kernels of the diagnostics library (the same device functions on arrays): no v_fma_f64.
* the other NumPy-order site (the MCMC mixture, acq.hip) likewise.
The same probes are compared with NumPy bit for bit on the MI355X by tests/test_gpu_parity.py::test_stretch_move_bits.
"""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from robo_amd import build as hip_build  # noqa: E402

CSRC = os.path.join(ROOT, "robo_amd", "csrc")
# the product's code-generation flags (everything of build.FLAGS that is not about linking)
GEN_FLAGS = [f for f in hip_build.FLAGS if f not in ("-fPIC", "-shared")]

pytestmark = pytest.mark.skipif(not os.path.exists(hip_build.HIPCC), reason="no hipcc")


def _emit(src, kind, tmp_path, extra=()):
    out = str(tmp_path / (os.path.basename(src) + (".ll" if kind == "ll" else ".s")))
    cmd = [hip_build.HIPCC] + GEN_FLAGS + list(extra) + ["--cuda-device-only", "-S"] + (["-emit-llvm"] if kind == "ll" else []) + \
        [os.path.join(CSRC, src), "-o", out]
    subprocess.run(cmd, check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    return out


def _ir_functions(path):
    """{kernel name: [lines]} of an LLVM assembly file"""
    out, cur = {}, None
    for line in open(path):
        if line.startswith("define "):
            cur = re.search(r"@([\w.$]+)\(", line).group(1)
            out[cur] = []
        elif line.startswith("}"):
            cur = None
        elif cur is not None:
            out[cur].append(line.rstrip("\n"))
    return out


_BIN = re.compile(r"^\s*(%[\w.]+) = (fmul|fadd|fsub|fdiv)((?: [a-z]+)*) double ([^,]+), (.+?)\s*$")


def _defs(lines):
    d = {}
    for ln in lines:
        m = _BIN.match(ln)
        if m:
            d[m.group(1)] = (m.group(2), m.group(3).split(), m.group(4).strip(), m.group(5).strip())
    return d


def _stretch_q_sites(lines):
    """every  q = fsub c, (fmul z, (fsub c, s))  dataflow in a function body -> list of the three flag lists"""
    d = _defs(lines)
    sites = []
    for q, (op, fl, a, b) in d.items():
        if op != "fsub" or b not in d:
            continue
        mop, mfl, ma, mb = d[b]
        if mop != "fmul":
            continue
        for t in (ma, mb):
            if t in d and d[t][0] == "fsub" and d[t][2] == a:
                sites.append((fl, mfl, d[t][1]))
    return sites


def _stretch_z_sites(lines):
    """every  z = fdiv (fmul t, t), a  with  t = fadd (fmul (fadd a, -1), u), 1"""
    d = _defs(lines)
    sites = []
    for z, (op, fl, num, den) in d.items():
        if op != "fdiv" or num not in d:
            continue
        sop, sfl, sa, sb = d[num]
        if sop != "fmul" or sa != sb or sa not in d:
            continue
        top, tfl, ta, tb = d[sa]
        if top != "fadd" or not tb.startswith("1.0") or ta not in d:
            continue
        mop, mfl, mx, my = d[ta]
        if mop == "fmul":
            sites.append((fl, sfl, tfl, mfl))
    return sites


def _no_fusable(flag_lists):
    return all("contract" not in fl and "fast" not in fl and "reassoc" not in fl for fl in flag_lists)


@pytest.fixture(scope="module")
def ir(tmp_path_factory):
    tmp = tmp_path_factory.mktemp("isa")
    return {src: _ir_functions(_emit(src, "ll", tmp)) for src in ("mcmc.hip", "potrf.hip")}


def test_stretch_move_is_not_contractable_in_the_chain_kernels(ir):
    kernels = {k: v for src in ir.values() for k, v in src.items()
               if "mcmc_propose_scale_kernel" in k or "mcmc_block_step_kernel" in k or "mcmc_block2_step_kernel" in k}
    # the launch-per-phase proposal + four one-block instantiations + two two-block ones
    assert len(kernels) == 7, sorted(kernels)
    for name, lines in kernels.items():
        q = _stretch_q_sites(lines)
        z = _stretch_z_sites(lines)
        assert len(q) == 1, (name, q)
        assert len(z) == 1, (name, z)
        assert _no_fusable(q[0]), (name, q)
        assert _no_fusable(z[0]), (name, z)


def test_the_check_sees_a_contractable_proposal(tmp_path):
    """the detector itself: the round-5 form of the proposal (plain operators) IS flagged"""
    src = tmp_path / "old_form.hip"
    src.write_text(
        '#include <hip/hip_runtime.h>\n'
        '__global__ void old_q(const double* c, const double* s, const double* z, double* q) {\n'
        '    const int i = threadIdx.x;\n'
        '    q[i] = __dsub_rn(c[i], __dmul_rn(z[i], __dsub_rn(c[i], s[i])));\n'
        '}\n')
    out = str(tmp_path / "old_form.ll")
    subprocess.run([hip_build.HIPCC] + GEN_FLAGS + ["--cuda-device-only", "-S", "-emit-llvm", str(src), "-o", out],
                   check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    fn = _ir_functions(out)
    (lines,) = fn.values()
    sites = _stretch_q_sites(lines)
    assert len(sites) == 1 and not _no_fusable(sites[0]), sites
    asm = str(tmp_path / "old_form.s")
    subprocess.run([hip_build.HIPCC] + GEN_FLAGS + ["--cuda-device-only", "-S", str(src), "-o", asm],
                   check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    from isa_count import kernels
    (ins,) = kernels(asm).values()
    assert any(i.startswith("v_fma_f64") for i in ins)       # what the MI355X executed in round 5


def test_accept_statistic_is_not_contractable(ir):
    for src, name in (("mcmc.hip", "mcmc_accept_kernel"), ("mcmc.hip", "mcmc_tail_kernel"), ("potrf.hip", "mcmc_block_step_kernel"),
                      ("potrf.hip", "mcmc_block2_step_kernel")):
        for k, lines in ir[src].items():
            if name not in k:
                continue
            d = _defs(lines)
            # lnpdiff = fsub (fadd (fmul (P - 1), log z), lp_new), lp_old
            sites = []
            for v, (op, fl, a, b) in d.items():
                if op == "fsub" and a in d and d[a][0] == "fadd" and d[a][2] in d and d[d[a][2]][0] == "fmul":
                    inner = d[d[a][2]]
                    if not fl and not d[a][1] and not inner[1]:
                        sites.append(v)
            assert sites, "no unfused accept statistic found in %s" % k


def test_probe_kernels_machine_code(tmp_path):
    from collections import Counter
    from isa_count import kernels
    ks = kernels(_emit(os.path.join("diag", "selftest.hip"), "s", tmp_path))
    seen = 0
    for name, ins in ks.items():
        c = Counter(i for i in ins if i.startswith("v_") and "f64" in i)
        if "stretch_q_probe_kernel" in name:
            assert c == {"v_add_f64": 2, "v_mul_f64": 1}, c
            seen += 1
        elif "lnpdiff_probe_kernel" in name:
            assert not any("fma" in k for k in c) and c["v_mul_f64"] == 1 and c["v_add_f64"] == 2 + 1, c   # (+ P - 1.0)
            seen += 1
        elif "stretch_z_probe_kernel" in name:
            # (a - 1), * u, + 1, t * t: two adds, two multiplies; the IEEE division expansion brings its own
            # v_div_scale / v_rcp / v_fma / v_mul / v_div_fmas / v_div_fixup
            assert c["v_add_f64"] == 2 and c["v_mul_f64"] == 3 and c["v_div_fixup_f64"] == 1, c
            seen += 1
    assert seen == 3


def test_mixture_kernel_squares_before_adding(tmp_path):
    """acq.hip mixture_kernel: NumPy's var is mean(multiply(d, d)) -- d * d rounded, then added"""
    fn = _ir_functions(_emit("acq.hip", "ll", tmp_path))
    (lines,) = [v for k, v in fn.items() if "mixture_kernel" in k]
    d = _defs(lines)
    squares = [v for v, (op, fl, a, b) in d.items() if op == "fmul" and a == b]
    assert squares
    assert all(not d[v][1] for v in squares), [(v, d[v]) for v in squares]
    assert not any("llvm.fma" in ln or "llvm.fmuladd" in ln for ln in lines)

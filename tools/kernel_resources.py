"""Per-kernel register / LDS / scratch table of the product's HIP sources, from the compiler's own resource remarks
(hipcc -Rpass-analysis=kernel-resource-usage, the product's flags, gfx950).  Needs no GPU.

    python tools/kernel_resources.py > profiles/r05z_kernel_resources.txt

What to read in it: ScratchSize / spills (0 everywhere is the goal: a spilled accumulator re-reads through HBM-backed
scratch inside the k-loop), VGPRs + AGPRs against the 512-register budget that fixes waves/SIMD, LDS per workgroup
against 160 KB per CU (static LDS only -- dynamic LDS is chosen at launch: potrf.hip / trsm.hip state theirs).
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from robo_amd import build as B      # noqa: E402

KEYS = ["VGPRs", "AGPRs", "TotalSGPRs", "ScratchSize [bytes/lane]", "VGPRs Spill", "SGPRs Spill", "Occupancy [waves/SIMD]",
        "LDS Size [bytes/block]"]
HEAD = ["VGPR", "AGPR", "SGPR", "scratch B/lane", "VGPR spill", "SGPR spill", "waves/SIMD", "static LDS B"]


def demangle(names):
    out = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.split("\n")
    return [re.sub(r"\(.*", "", o).replace("robo::", "").replace("void ", "") for o in out[:len(names)]]


def main():
    flags = [f for f in B.FLAGS if f not in ("-shared", "-fPIC")]
    rows = []
    with tempfile.TemporaryDirectory() as tmp:
        for src in B.sources() + B.diag_sources():
            cmd = [B.HIPCC] + flags + ["--cuda-device-only", "-c", "-I", os.path.join(ROOT, "include"),
                                       "-Rpass-analysis=kernel-resource-usage", src, "-o", os.path.join(tmp, "x.o")]
            err = subprocess.run(cmd, capture_output=True, text=True).stderr
            cur = None
            for line in err.split("\n"):
                m = re.search(r"remark:\s+Function Name: (\S+)", line)
                if m:
                    cur = {"file": os.path.relpath(src, ROOT), "name": m.group(1)}
                    rows.append(cur)
                    continue
                m = re.search(r"remark:\s+([A-Za-z \[\]/]+): (\S+) \[-Rpass", line)
                if m and cur is not None:
                    cur[m.group(1).strip()] = m.group(2)
    names = demangle([r["name"] for r in rows])
    head = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
    print("kernel resources at %s: hipcc %s, -Rpass-analysis=kernel-resource-usage" % (head, " ".join(flags)))
    print("%-34s %-58s" % ("source", "kernel") + " ".join("%14s" % h for h in HEAD))
    spills = 0
    for r, n in zip(rows, names):
        print("%-34s %-58s" % (r["file"], n[:58]) + " ".join("%14s" % r.get(k, "-") for k in KEYS))
        spills += int(r.get("ScratchSize [bytes/lane]", "0") or 0) > 0
    print("%d kernels, %d with scratch" % (len(rows), spills))


if __name__ == "__main__":
    main()

"""Instruction mix of device kernels from hipcc's -save-temps assembly: total, VALU, fp64 VALU, MFMA, LDS, global, barriers.
usage: python tools/isa_count.py <file.s> <kernel-name-substring> [...]   (static counts; loops are counted once)"""
import sys
from collections import Counter


def kernels(path):
    cur, out = None, {}
    for line in open(path):
        if line.startswith("_Z") and ":" in line:
            cur = line.split(":")[0]
            out[cur] = []
        elif line.startswith(".Lfunc_end"):
            cur = None
        elif cur and line.startswith("\t"):
            t = line.strip()
            if t and not t.startswith((".", ";")):
                out[cur].append(t.split()[0])
    return out


if __name__ == "__main__":
    ks = kernels(sys.argv[1])
    for name, ins in ks.items():
        if not any(p in name for p in sys.argv[2:]):
            continue
        c = Counter(ins)
        def n(pred):
            return sum(v for k, v in c.items() if pred(k))
        print("%s\n  total %d  VALU %d (fp64 %d)  MFMA %d  LDS %d  global %d  s_barrier %d  s_waitcnt %d" % (
            name, len(ins), n(lambda k: k.startswith("v_") and not k.startswith("v_mfma")),
            n(lambda k: k.startswith("v_") and "f64" in k and not k.startswith("v_mfma")),
            n(lambda k: k.startswith("v_mfma")), n(lambda k: k.startswith("ds_")), n(lambda k: k.startswith("global_")),
            c.get("s_barrier", 0), c.get("s_waitcnt", 0)))
        print("  most frequent:", ", ".join("%s x%d" % kv for kv in c.most_common(12)))

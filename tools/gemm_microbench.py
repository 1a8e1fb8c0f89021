import sys; sys.path.insert(0,'.')
from robo_amd import _lib
ctx=_lib.default_context()
for K in (512, 2048, 4096):
    for v in (0,1):
        for wgs in (512, 1024):
            print("K",K,"variant",v,"wgs",wgs, ctx.microbench_gemm_f64(v, wgs, K, 5), flush=True)

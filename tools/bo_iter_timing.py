"""One BO iteration at the headline size with the reference's defaults (bench.bo_iteration), with and without the asynchronous
W = L^-1 build that GaussianProcess.train() starts after its final fit (robo_gp_prefetch_inverse)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from robo_amd import _lib
for N, D in ((4096, 16), (2048, 16), (1024, 8)):
    with_pf = bench.bo_iteration(N, D)["ms"]
    orig = _lib.DeviceGP.prefetch_inverse
    _lib.DeviceGP.prefetch_inverse = lambda self: None
    try:
        without = bench.bo_iteration(N, D)["ms"]
    finally:
        _lib.DeviceGP.prefetch_inverse = orig
    print("N=%d D=%d  with prefetch: train %.3f maximize %.3f total %.3f ms   without: train %.3f maximize %.3f total %.3f ms"
          % (N, D, with_pf["train"], with_pf["maximize"], with_pf["total"], without["train"], without["maximize"], without["total"]))

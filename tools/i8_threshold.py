"""Where does the int8 path pay?  Per-global-step time of the stepping API, fp64 matrix cores vs int8 x 6, many chains, various D.
Run on the GPU box: python tools/i8_threshold.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from riemannhamiltonianmontecarlo_amd import _capi
from riemannhamiltonianmontecarlo_amd.data import synthetic_logreg
lib = _capi.load_hip_library()
SHAPES = [(690, 15, 8192), (1000, 25, 8192), (2000, 33, 8192), (5000, 48, 8192), (10000, 64, 2048), (10000, 64, 1024), (10000, 32, 8192), (10000, 64, 512), (10000, 64, 256), (10000, 64, 128), (1000, 25, 600)]
if len(sys.argv) > 1:
    SHAPES = [(10000, 64, 512), (10000, 64, 256), (10000, 64, 128), (1000, 25, 600), (10000, 64, 8192)]
for (M, D, n) in SHAPES:
    XX, t = synthetic_logreg(M, D, 1)
    res = []
    for fl in (0, _capi.int8_metric_flags(6)):
        with lib.context(M, D, n, flags=fl) as ctx:
            ctx.set_data(XX, t)
            ctx.chains_init(seed=1, L=6, eps=0.3, K=4)
            ctx.chains_run(3)
            t0 = time.perf_counter(); ctx.chains_run(10); t1 = time.perf_counter()
            res.append((t1 - t0) / 10 * 1e3)
    print("M=%5d D=%2d chains=%5d: fp64 %.3f ms/step, int8x6 %.3f ms/step  (ratio %.2f)" % (M, D, n, res[0], res[1], res[0] / res[1]))

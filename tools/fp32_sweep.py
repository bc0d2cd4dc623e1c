#!/usr/bin/env python3
"""BASELINE config 5 asks for an "fp32 vs fp64 tolerance sweep".  With RMHMC_FLAG_FP32_METRIC the metric assemblies
X' diag(v) X run on the fp32 matrix cores (f32 operands and accumulators), everything else stays float64.  This
script measures, against the float64 path on the same inputs, the relative error on theta and on log|G| after ONE
leapfrog step (the north_star parity statement: 1e-6) and the time of a leapfrog step in both modes.
Run on the GPU box:  python tools/fp32_sweep.py > gpurun_out/fp32_sweep.txt"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401
from riemannhamiltonianmontecarlo_amd import _capi  # noqa: E402
from riemannhamiltonianmontecarlo_amd.data import synthetic_logreg  # noqa: E402

lib = _capi.load_hip_library()
print("%8s %5s %7s | %12s %12s %12s | %10s %10s" % ("M", "D", "chains", "err theta", "err log|G|", "err p", "ms f64", "ms f32"))
for M, D, n in ((1000, 8, 1024), (10000, 16, 1024), (10000, 64, 2048), (50000, 64, 1024), (20000, 128, 512), (50000, 256, 512)):
    XX, t = synthetic_logreg(M, D, 0)
    rs = np.random.RandomState(1)
    w = 0.05 * rs.randn(n, D); p = np.sqrt(M / 4.0) * rs.randn(n, D)   # momentum of the metric's scale
    out, ms = [], []
    for flags in (0, _capi.FLAG_FP32_METRIC):
        with lib.context(M, D, n, flags=flags, options={"fused": 0}) as ctx:   # the D <= 8 fused kernel has no fp32 mode: compare the generic kernels
            ctx.set_data(XX, t)
            out.append(ctx.leapfrog(w, p, 0.5, 1, 1, 4))
            ctx.chains_init(theta0=w, seed=1)
            ctx.chains_run(1)
            t0 = time.perf_counter(); ctx.chains_run(3); ms.append((time.perf_counter() - t0) / 3 * 1e3)
    (w64, p64, h64, _), (w32, p32, h32, _) = out
    rel = lambda a, b: float(np.max(np.abs(a - b)) / np.max(np.abs(b)))
    print("%8d %5d %7d | %12.3e %12.3e %12.3e | %10.2f %10.2f" % (M, D, n, rel(w32, w64), rel(2 * h32, 2 * h64), rel(p32, p64), ms[0], ms[1]))

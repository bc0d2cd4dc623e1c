#!/usr/bin/env python3
"""Where does the fused small-problem kernel spend its time?  Per global step for 1024 chains, by M and K."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa
from riemannhamiltonianmontecarlo_amd import _capi
from riemannhamiltonianmontecarlo_amd.data import synthetic_logreg
lib = _capi.load_hip_library()
for D in (8, 4):
    for M in (64, 256, 1024, 1536):
        for K in (1, 2, 4):
            XX, t = synthetic_logreg(M, D, 0)
            with lib.context(M, D, 1024, flags=0) as ctx:
                ctx.set_data(XX, t)
                ctx.chains_init(seed=1, K=K)
                ctx.chains_run(20)
                t0 = time.perf_counter(); ctx.chains_run(200); dt = (time.perf_counter() - t0) / 200
            print("D=%d M=%5d K=%d  %.1f us/step" % (D, M, K, dt * 1e6))

"""The drop-in shim at BASELINE config 3 as a user would call it: RMHMC(XX, t, 300, 100, n_chains=8192, verbose=True).
Run on the GPU box: python tools/shim_c3_run.py   (round 2: TimeTaken 16.6 s for 199 post-burn-in transitions of 8192 chains = 344 k leapfrog-steps/s,
the reference's progress lines on stdout)"""
import sys, time
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
from riemannhamiltonianmontecarlo_amd import RMHMC
from riemannhamiltonianmontecarlo_amd.data import synthetic_logreg
XX, t = synthetic_logreg(10000, 64, 0)
t0 = time.perf_counter()
smp, secs, info = RMHMC(XX, t, 300, 100, n_chains=8192, seed=7, compat=False, verbose=True, return_info=True)
wall = time.perf_counter() - t0
print("shape", smp.shape, "TimeTaken %.2f s, wall %.2f s, post-burn-in leapfrog steps/s %.0f, acceptance %.3f" % (secs, wall, info["leapfrog_steps"].sum() / secs, info["accepted"].mean() / 300))

"""Plain HMC for many chains on the bundled australian data: whole-trajectory kernel vs the generic five-launches-per-step path.
Run on the GPU box: python tools/bench_hmc_batch.py  (option hmc_traj_maxn picks the path)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from riemannhamiltonianmontecarlo_amd import HMC
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
d = np.load(os.path.join(GOLDEN, "data_australian.npz"))
for n in (512, 2048, 8192):
  for name, opts in (("one launch per trajectory", {"hmc_traj_maxn": 100000}), ("generic", {"hmc_traj_maxn": 0})):
    w, secs, info = HMC(d["XX"], d["t"], 120, 20, 100, 0.02, n_chains=n, seed=3, verbose=False, return_info=True, options=opts)
    steps = int(info["leapfrog_steps"].sum())
    print(name, "chains %5d: %.3f s post burn-in, %.2f M leapfrog-steps/s, acceptance %.3f" % (n, secs, steps * (100 / 120.0) / secs / 1e6, float(info["accepted"].sum()) / (120 * n)))

"""Plain HMC exactly as the reference's main.py calls it (one chain, 100 leapfrog steps max, eps 0.14) on the bundled data sets:
seconds of the post-burn-in phase.  Run on the GPU box: python tools/bench_hmc_single.py [iterations]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from riemannhamiltonianmontecarlo_amd import HMC
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
it = int(sys.argv[1]) if len(sys.argv) > 1 else 1200
for name in ("australian", "german", "heart", "pima", "ripley"):
    d = np.load(os.path.join(GOLDEN, "data_%s.npz" % name))
    w, secs, info = HMC(d["XX"], d["t"], it, it // 6, seed=3, verbose=False, return_info=True)
    steps = int(info["leapfrog_steps"].sum())
    print("%-10s D=%2d: %d iterations, %d leapfrog steps, %.3f s post burn-in, acceptance %.3f, mean |w| %.3f" % (
        name, d["XX"].shape[1], it, steps, secs, float(info["accepted"].sum()) / it, float(np.abs(w.mean(0)).mean())))

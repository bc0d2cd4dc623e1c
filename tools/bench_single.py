"""Single-chain latency of the stepping API on the bundled data sets (what the reference's main.py runs), run on the GPU box:
python tools/bench_single.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from riemannhamiltonianmontecarlo_amd import _capi
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
lib = _capi.load_hip_library()
for name in ("australian", "german", "heart", "pima", "ripley"):
    d = np.load(os.path.join(GOLDEN, "data_%s.npz" % name))
    XX, t = d["XX"], d["t"].reshape(-1)
    for n in (1, 10, 64):
        with lib.context(XX.shape[0], XX.shape[1], n, flags=0) as ctx:
            ctx.set_data(XX, t)
            ctx.chains_init(seed=1, L=6, eps=0.5, K=4)
            ctx.chains_run(50)
            t0 = time.perf_counter(); ctx.chains_run(400); t1 = time.perf_counter()
            print("%-10s M=%4d D=%2d chains=%3d: %7.1f us per global step, %9.0f leapfrog-steps/s" % (
                name, XX.shape[0], XX.shape[1], n, (t1 - t0) / 400 * 1e6, n * 400 / (t1 - t0)))

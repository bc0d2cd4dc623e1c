import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa
from riemannhamiltonianmontecarlo_amd import _capi
lib = _capi.load_hip_library()
d = np.load(os.path.join(ROOT, "tests", "golden", "data_australian.npz"))
for n in (1, 10, 1000):
    with lib.context(d["XX"].shape[0], d["XX"].shape[1], n, flags=0) as ctx:
        ctx.set_data(d["XX"], d["t"])
        ctx.chains_init(seed=1)
        ctx.chains_run(20)
        t0 = time.perf_counter(); ctx.chains_run(300); dt = (time.perf_counter() - t0) / 300
    print("RMHMC_GRAPH=%s australian n=%4d: %.1f us per global step" % (os.environ.get("RMHMC_GRAPH", "1"), n, dt * 1e6))

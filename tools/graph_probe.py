import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa
from riemannhamiltonianmontecarlo_amd import _capi
lib = _capi.load_hip_library()
d = np.load(os.path.join(ROOT, "tests", "golden", "data_australian.npz"))
for n, graph in ((n, g) for n in (1, 10, 1000) for g in (1, 0)):
    with lib.context(d["XX"].shape[0], d["XX"].shape[1], n, flags=0, options={"medium": 0}) as ctx:   # (generic path: ~40 launches per step)
        ctx.set_data(d["XX"], d["t"])
        ctx.set_option("graph", graph)
        ctx.chains_init(seed=1)
        ctx.chains_run(20)
        t0 = time.perf_counter(); ctx.chains_run(300); dt = (time.perf_counter() - t0) / 300
    print("graph=%d australian n=%4d: %.1f us per global step" % (graph, n, dt * 1e6))

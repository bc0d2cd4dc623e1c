"""Is the fp64 generic path monotone in the batch size?  Per-global-step time, D 64, M 10000, fp64 matrix cores only, with the per-kernel
split for the small batches.  Run on the GPU box: python tools/fp64_batch_sweep.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from riemannhamiltonianmontecarlo_amd import _capi
from riemannhamiltonianmontecarlo_amd.data import synthetic_logreg
lib = _capi.load_hip_library()
M, D = 10000, 64
XX, t = synthetic_logreg(M, D, 1)
for n in (64, 128, 256, 512, 1024, 2048, 4096, 8192):
    with lib.context(M, D, n, flags=0) as ctx:
        ctx.set_data(XX, t)
        ctx.chains_init(seed=1, L=6, eps=0.3, K=4)
        ctx.chains_run(3)
        t0 = time.perf_counter(); ctx.chains_run(10); t1 = time.perf_counter()
        ctx.kernel_time("enable"); ctx.kernel_time("reset"); ctx.chains_run(4)
        kt = {k: round(ctx.kernel_time(k)[0] / 4 * 1e3, 2) for k in ("assemble", "leverage", "rowpass", "mompass", "factor", "small")}
    print("chains=%5d: %.3f ms/step  %.0f steps/s  %s" % (n, (t1 - t0) / 10 * 1e3, n * 10 / (t1 - t0), kt))

"""Time the metric assembly of ONE point evaluation at config 3's shape on REAL operand planes (v of a random position, the data's
x_a x_b), with the library given in RMHMC_HIP_LIB - the normal build or a -DI8_ABLATE=n timing build (results meaningless) - to see
what the MFMA stream alone sustains on these bytes (tools/i8_gemm_probe uses random bytes).  Run on the GPU box."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
from riemannhamiltonianmontecarlo_amd import _capi
from riemannhamiltonianmontecarlo_amd.data import synthetic_logreg
M, D, n = 10000, 64, 8192
XX, t = synthetic_logreg(M, D, 0)
rs = np.random.RandomState(1)
lib = _capi.load_hip_library()
for S in (6, 5, 4):
    with lib.context(M, D, n, flags=_capi.int8_metric_flags(S)) as ctx:
        ctx.set_data(XX, t)
        w = 0.05 * rs.randn(n, D)
        ctx.log_posterior(w)                      # warm-up evaluation
        ctx.kernel_time("enable"); ctx.kernel_time("reset")
        for _ in range(3):
            ctx.log_posterior(w)
        s, k = ctx.kernel_time("assemble_i8")
        s2, k2 = ctx.kernel_time("leverage_i8")
        ops = 2.0 * n * M * (D * (D + 1) // 2) * (S * (S + 1) // 2)
        print("%s S=%d assemble_i8 %.3f ms/launch (%d) = %.2f POP/s; leverage_i8 %.3f ms" % (os.path.basename(os.environ.get("RMHMC_HIP_LIB", "lib")), S, s / k * 1e3, k, ops / (s / k) / 1e15, s2 / max(1, k2) * 1e3), flush=True)

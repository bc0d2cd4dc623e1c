"""Diagnostic: int8-sliced metric assembly vs the fp64 matrix-core assembly and vs the oracle, S = 4..7.
Run on the GPU box:  python tools/i8_check.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from riemannhamiltonianmontecarlo_amd import _capi
from riemannhamiltonianmontecarlo_amd.data import synthetic_logreg

hip = _capi.load_hip_library()
orc = _capi.RmhmcLib(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle", "librmhmc_oracle.so"))


def rel(a, b):
    return float(np.abs(a - b).max() / max(1e-300, np.abs(b).max()))


for (M, D, n) in [(2000, 64, 200), (1000, 25, 130), (532, 8, 40), (10000, 64, 256)]:
    XX, t = synthetic_logreg(M, D, 3)
    rs = np.random.RandomState(1)
    w = 0.4 * rs.randn(n, D) / np.sqrt(D)
    p = rs.randn(n, D) * 3
    with orc.context(M, D, n, flags=0) as ctx:
        ctx.set_data(XX, t)
        Go, hldo, go = ctx.metric(w)
        lo = ctx.leapfrog(w, p, 0.5, 1, 1, K=4)
    for S in (0, 4, 5, 6, 7):
        fl = _capi.int8_metric_flags(S) if S else 0
        with hip.context(M, D, n, flags=fl) as ctx:
            ctx.set_data(XX, t)
            G, hld, g = ctx.metric(w)
            lf = ctx.leapfrog(w, p, 0.5, 1, 1, K=4)
        eG = max(rel(G[c], Go[c]) for c in range(n))
        print("M=%d D=%d n=%d S=%d: G %.2e  hld %.2e  | leapfrog theta %.2e  p %.2e  logdet %.2e" % (
            M, D, n, S, eG, float(np.abs(hld - hldo).max()), max(rel(lf[0][c], lo[0][c]) for c in range(n)),
            max(rel(lf[1][c], lo[1][c]) for c in range(n)), float(np.abs((lf[2] - lo[2]) / lo[2]).max())))

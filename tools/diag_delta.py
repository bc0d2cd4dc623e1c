"""Diagnostic: delta assembly vs full assembly vs oracle, per number of leapfrog steps (run on the GPU box)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import __graft_entry__ as ge
from riemannhamiltonianmontecarlo_amd import _capi
from riemannhamiltonianmontecarlo_amd.data import synthetic_logreg

def rel(a, b):
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))

hip = _capi.load_hip_library(); oracle = _capi.RmhmcLib(ge.ORACLE_LIB)
for (M, D, n) in [(129, 48, 300), (203, 33, 7), (900, 64, 2100), (400, 40, 2432), (10000, 64, 1024)]:
    XX, t = synthetic_logreg(M, D, 5)
    rs = np.random.RandomState(M + n)
    w = 0.4 * rs.randn(n, D) / np.sqrt(D); p = rs.randn(n, D)
    dirs = np.where(rs.rand(n) < 0.5, -1, 1).astype(np.int32)
    for ns in (1, 2, 3):
        res = {}
        for name, opts, fl in (("full", {"i8_delta": 0}, 0), ("delta", {"i8_delta": 1, "i8_delta_inner": 1}, 0),
                               ("delta_end", {"i8_delta": 1, "i8_delta_inner": 0}, 0),
                               ("full_innerfull", {"i8_delta": 0}, _capi.FLAG_INT8_INNER_FULL), ("fp64", {}, None)):
            flags = 0 if fl is None else (_capi.int8_metric_flags(6) | fl)
            with hip.context(M, D, n, flags=flags, options=opts or None) as ctx:
                ctx.set_data(XX, t, 100.0)
                res[name] = ctx.leapfrog(w, p, 0.5, dirs, ns, 4)
        nn = min(n, 64)
        with oracle.context(M, D, nn, flags=0) as ctx:
            ctx.set_data(XX, t, 100.0)
            ref = ctx.leapfrog(w[:nn], p[:nn], 0.5, dirs[:nn], ns, 4)
        def mx(a, b, m=n):
            return max(rel(a[0][c], b[0][c]) for c in range(m))
        print(M, D, n, "steps", ns, "delta_end-full %.1e delta-full %.1e  innerfull-full %.1e  fp64-full %.1e | vs oracle: full %.1e delta %.1e fp64 %.1e | max|theta| %.1e"
              % (mx(res["delta_end"], res["full"]), mx(res["delta"], res["full"]), mx(res["full_innerfull"], res["full"]), mx(res["fp64"], res["full"]),
                 mx(res["full"], ref, nn), mx(res["delta"], ref, nn), mx(res["fp64"], ref, nn), np.abs(res["full"][0]).max()), flush=True)

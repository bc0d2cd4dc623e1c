import sys, numpy as np
sys.path.insert(0, '.')
import torch
from riemannhamiltonianmontecarlo_amd import _capi
from riemannhamiltonianmontecarlo_amd.data import synthetic_logreg
hip = _capi.load_hip_library()
M, D, n = 200, 5, 4
XX, t = synthetic_logreg(M, D, 2)
def run(chunks):
    out = []
    with hip.context(M, D, n) as ctx:
        ctx.set_data(XX, t)
        ctx.chains_init(seed=9)
        for k in chunks:
            ctx.chains_run(k)
            w, it, a = ctx.chains_state()
            out.append((w.copy(), it.copy()))
    return out
a = run([1] * 12)
b = run([12])
c = run([3, 3, 3, 3])
d = run([12])
print("12x1 vs 1x12 :", np.abs(a[-1][0] - b[-1][0]).max(), a[-1][1], b[-1][1])
print("4x3 vs 1x12  :", np.abs(c[-1][0] - b[-1][0]).max())
print("1x12 vs 1x12 :", np.abs(d[-1][0] - b[-1][0]).max())
for k in range(1, 13):
    e = run([k]); f = run([1] * k)
    print(k, np.abs(e[-1][0] - f[-1][0]).max(), e[-1][1], f[-1][1])

"""NumPy restatement of the reference's LITERAL algorithm - TEST INFRASTRUCTURE, like everything under oracle/.

Only tests/ and bench.py's cpu_baseline leg may import this module; the product never does.

What it is for: north_star asks for the GPU numbers "next to the reference NumPy path timed on the host cores".  The
reference itself (code/rmhmc.py) cannot travel to the GPU box, so this module restates what it computes - the D x D x D
tensor InvGdG[d] = G^-1 dG/dw_d built per point (rmhmc.py:64-77,142-156, O(M D^3)), LU `inv` / `solve` exactly where the
reference uses them (:58,113,121,138), the naive log(1 + e^f) and e^f / (1 + e^f) (:100,167) - in NumPy, with the tensor formed by
one matrix product per block of data rows instead of the reference's Python double loop and D products, so the arithmetic that is
timed is NumPy / BLAS / LAPACK like the reference's, minus interpreter overhead (it is therefore FASTER than rmhmc.py - 0.16 s per
leapfrog step at config 3's shape on 8 cores where SURVEY measured ~0.9 s for the reference - and conservative as a baseline).  Pinned to the same golden tapes as the C oracle
(tests/test_oracle_numpy.py: every captured transition of the small tapes to 1e-8).

Citations are to /root/reference/code/rmhmc.py.
"""
import numpy as np

ALPHA = 100.0  # rmhmc.py:19


class Point:
    """Everything the sampler keeps at a position w: the set-up block rmhmc.py:50-77 / the end-of-step block :134-156."""

    def __init__(self, X, t, w, alpha=ALPHA, rows_per_block=2048):
        M, D = X.shape
        with np.errstate(all="ignore"):
            f = X @ w
            p = 1.0 / (1.0 + np.exp(-f))                      # :52
            v = p * (1.0 - p)                                 # :53
            self.G = (X.T * v) @ X + np.eye(D) / alpha        # :57
            self.Ginv = np.linalg.inv(self.G)                 # :58 (LU)
            ef = np.exp(f)
            self.grad = X.T @ (t - ef / (1.0 + ef)) - w / alpha   # :100,140
            c = v * (1.0 - 2.0 * p)                           # :67-69
            # dG[d] = X' diag(c x_d) X for every d (:66-75) as ONE matrix product per block of data rows: rows (d, a) of the left factor
            # are c_n x_nd x_na (the reference builds the same products column by column, :73-75, and multiplies per d)
            dG = np.zeros((D * D, D))
            for lo in range(0, M, rows_per_block):
                Xb = X[lo:lo + rows_per_block]
                W = (c[lo:lo + rows_per_block, None, None] * Xb[:, :, None] * Xb[:, None, :]).reshape(Xb.shape[0], D * D)
                dG += W.T @ Xb
            dG = dG.reshape(D, D, D)
            self.T = self.Ginv @ dG                            # InvGdG[d] = InvG.dot(GDeriv), :76
            self.tr = np.trace(self.T, axis1=1, axis2=2)       # :77
        self.w = w

    def log_joint(self, X, t, alpha=ALPHA):
        with np.errstate(all="ignore"):
            f = X @ self.w
            D = self.w.size
            prior = np.sum(-0.5 * np.log(2.0 * np.pi * alpha) - self.w ** 2 / (2.0 * alpha))   # tools.py:10-14
            return f @ t - np.sum(np.log(1.0 + np.exp(f))) + prior                           # :166-169

    def half_logdet(self):
        return float(np.sum(np.log(np.diag(np.linalg.cholesky(self.G)))))                      # :171


def leapfrog(X, t, pt, p, eps, tau, K, guards=True, alpha=ALPHA):
    """One generalised leapfrog step from the point record pt with momentum p (rmhmc.py:96-163); returns (new record, p)."""
    h = tau * eps / 2.0
    pm = p.copy()
    for _ in range(K):                                          # :103-108
        u = pt.Ginv @ pm
        last = 0.5 * np.einsum("a,dab,b->d", pm, pt.T, u)       # 0.5 PM' InvGdG[d] InvG PM
        pm = p + h * (pt.grad - 0.5 * pt.tr + last)
    p = pm
    with np.errstate(all="ignore"):
        u0 = np.linalg.solve(pt.G, p)                           # :113
        pw = pt.w.copy()
        M, D = X.shape
        for _ in range(K):                                      # :115-122
            f = X @ pw
            pr = 1.0 / (1.0 + np.exp(-f))
            v = pr * (1.0 - pr)
            Gk = (X.T * v) @ X + np.eye(D) / alpha
            pw = pt.w + h * (u0 + np.linalg.solve(Gk, p))
    if guards and np.linalg.norm(pw) > 10.0:                    # :125-130
        pw = pw / (np.linalg.norm(pw) * 3.0)
    new = Point(X, t, pw, alpha)                                # :134-156
    u = new.Ginv @ p
    last = 0.5 * np.einsum("a,dab,b->d", p, new.T, u)           # :158-161
    return new, p + h * (new.grad - 0.5 * new.tr + last)        # :163


def transition(X, t, w, z, u_len, g_dir, u_acc, L=6, eps=0.5, K=4, compat=True, alpha=ALPHA):
    """One MCMC transition with the reference's draws in the reference's order (rmhmc.py:80,89,90,181)."""
    cur = Point(X, t, w, alpha)
    Lc = np.linalg.cholesky(cur.G)
    p = Lc.T @ z if compat else Lc @ z                          # :80 (L' z in the Python reference)
    if compat and np.linalg.norm(p) > 100.0:                    # :81-85
        p = p / (np.linalg.norm(p) * 25.0)
    p0 = p.copy()
    nsteps = int(np.ceil(u_len * L))                            # :89
    tau = 1.0 if g_dir > 0.5 else -1.0                          # :90-93
    pt = cur
    for _ in range(nsteps):
        pt, p = leapfrog(X, t, pt, p, eps, tau, K, guards=compat, alpha=alpha)
    with np.errstate(all="ignore"):
        H_prop = -pt.log_joint(X, t, alpha) + pt.half_logdet() + 0.5 * p @ pt.Ginv @ p      # :166-172
        H_cur = -cur.log_joint(X, t, alpha) + cur.half_logdet() + 0.5 * p0 @ cur.Ginv @ p0  # :175-176
        ratio = H_cur - H_prop
        accept = bool(ratio > 0 or ratio > np.log(u_acc))       # :181
    return dict(w=pt.w if accept else w, accepted=accept, nsteps=nsteps, w_prop=pt.w, p_prop=p, H_prop=H_prop, H_cur=H_cur,
                hld_prop=pt.half_logdet())

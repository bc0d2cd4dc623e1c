"""The int8 (fixed-point) metric path where fixed point can hurt: badly scaled columns with an intercept, chains saturated on
every data row, one outlier row.  Each test asserts the DOCUMENTED bound of csrc/metric_i8.hip.h ("Error bound"):
    |dG_ab| <= S M 2^(e_ab - 8S),   2^e_ab > max_n |x_na x_nb|   (worst case, any chain state)
and that the certificate evaluated by rmhmc_set_data (RMHMC_FLAG_INT8_CERTIFY, set by the Python shims whenever they choose the
int8 path themselves) sends data it cannot certify to 1e-9 to the fp64 kernels.  The oracle is the fp64 CPU restatement of
rmhmc.py:51-58,64-77,96-163.  Needs an MI355X: run with  pytest -m gpu."""
import numpy as np
import pytest

from conftest import rel_err
from riemannhamiltonianmontecarlo_amd import _capi
from riemannhamiltonianmontecarlo_amd.data import synthetic_logreg

pytestmark = pytest.mark.gpu
S = 6


def _bound_matrix(XX, S):
    """S M 2^(e_ab - 8S) per column pair, e_ab = frexp exponent of max_n |x_na x_nb| (what k_zmax computes)."""
    M, D = XX.shape
    mx = np.zeros((D, D))
    for a in range(D):
        mx[a] = np.abs(XX[:, a:a + 1] * XX).max(axis=0)
    e = np.frexp(mx)[1]
    return S * M * np.ldexp(1.0, e - 8 * S)


def _both(hip, oracle, XX, t, n, fn, flags):
    M, D = XX.shape
    out = []
    for lib, fl in ((hip, flags), (oracle, 0)):
        with lib.context(M, D, n, flags=fl) as ctx:
            ctx.set_data(XX, t, 100.0)
            out.append((fn(ctx), ctx.int8_certificate()))
    return out


def _nan_rel(a, b):
    """rel_err over the finite entries; non-finite entries (a trajectory that diverges does so in the oracle too) must coincide"""
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    fa, fb = np.isfinite(a), np.isfinite(b)
    assert np.array_equal(fa, fb), "non-finite entries differ"
    if not fb.any():
        return 0.0
    return float(np.max(np.abs(a[fb] - b[fb])) / max(np.max(np.abs(b[fb])), 1e-300))


def _col_rel(a, b):
    """max over columns of (max_c |a - b|) / (max_c |b|): theta / p components live on the inverse column scales"""
    return float(np.max(np.abs(a - b).max(axis=0) / np.maximum(np.abs(b).max(axis=0), 1e-300)))


def test_badly_scaled_columns_and_intercept(hip, oracle):
    """Columns scaled by 1e+6 / 1e-6 / 1e+3 / 1e-3 next to an all-ones intercept.  The per-pair exponents follow the column scales, so
    the certificate stays tiny, the element-wise bound holds, and the SCALED metric S^-1 G S^-1 and one leapfrog step agree with the
    oracle as well as on N(0,1) data."""
    M, D, n = 4000, 24, 140
    X0, t = synthetic_logreg(M, D - 1, 3)
    sc = np.ones(D); sc[1] = 1e6; sc[2] = 1e-6; sc[3] = 1e3; sc[4] = 1e-3
    XX = np.hstack([np.ones((M, 1)), X0]) * sc
    rs = np.random.RandomState(0)
    w = (0.3 * rs.randn(n, D) / np.sqrt(D)) / sc
    p = rs.randn(n, D) * sc * 10.0

    def fn(ctx):
        return ctx.metric(w) + ctx.metric_terms(w, p) + ctx.leapfrog(w, p, 0.5, 1, 1, 4)

    (g, (bound, active)), (o, _) = _both(hip, oracle, XX, t, n, fn, _capi.int8_metric_flags(S) | _capi.FLAG_INT8_CERTIFY)
    assert active and bound < 1e-10, bound          # certified: scaling alone costs nothing
    Gg, hg, gg, trg, qg, wg, pg, h1g, sg = g
    Go, ho, go, tro, qo, wo, po, h1o, so = o
    B = _bound_matrix(XX, S)
    assert (np.abs(Gg - Go) <= B[None] + 1e-15 * np.abs(Go)).all()          # the documented worst-case bound, element by element
    scale = np.sqrt(np.einsum("cdd->cd", Go))
    Gs_g = Gg / (scale[:, :, None] * scale[:, None, :]); Gs_o = Go / (scale[:, :, None] * scale[:, None, :])
    assert np.abs(Gs_g - Gs_o).max() < 1e-12
    assert np.abs(hg - ho).max() < 1e-10 * np.abs(ho).max()
    assert _col_rel(trg, tro) < 1e-8 and _col_rel(qg, qo) < 1e-8
    assert _col_rel(wg, wo) < 1e-9 and _col_rel(pg, po) < 1e-9
    assert np.abs(h1g - h1o).max() < 1e-9 * np.abs(h1o).max()


@pytest.mark.parametrize("f0", [12.0, 30.0, 45.0])
def test_saturated_chains(hip, oracle, f0):
    """Every data row has |x_n.w| ~ f0 (intercept weight f0): v ~ e^-f0 sits near (12: 6e-6), at (30: 9e-14) or below (45: 3e-20) the
    absolute grid 2^-48 of v, so G ~ I/alpha + small.  The documented bound for this regime is |dG| alpha <= S M 2^(e-8S) alpha; theta
    after a step must still agree with the oracle to 1e-9 (typical error ~sqrt(M) below the bound)."""
    M, D, n = 10000, 32, 130
    X0, t = synthetic_logreg(M, D - 1, 1)
    XX = np.hstack([np.ones((M, 1)), X0])
    rs = np.random.RandomState(int(f0))
    w = 0.02 * rs.randn(n, D) / np.sqrt(D); w[:, 0] = f0 * np.where(rs.rand(n) < 0.5, -1.0, 1.0)
    p = 0.01 * rs.randn(n, D)

    def fn(ctx):
        return ctx.metric(w) + ctx.metric_terms(w, p) + ctx.leapfrog(w, p, 1e-4, 1, 1, 4)

    (g, (bound, active)), (o, _) = _both(hip, oracle, XX, t, n, fn, _capi.int8_metric_flags(S))
    Gg, hg, gg, trg, qg, wg, pg, h1g, sg = g
    Go, ho, go, tro, qo, wo, po, h1o, so = o
    assert np.abs(XX @ w.T).min() > 0.6 * f0                                   # saturated on every row
    B = _bound_matrix(XX, S)
    assert (np.abs(Gg - Go) <= B[None]).all()
    worst = float((B * 100.0).max())                                           # relative to G >= I/alpha
    assert worst < 1e-6
    assert np.isfinite(wo).all() and np.isfinite(po).all()
    for c in range(n):
        # the chain's own v grid (vexp, VSlice in kernels.hip.h) keeps a saturated chain at the accuracy of an ordinary one
        assert rel_err(Gg[c], Go[c]) < 1e-12, c
        assert rel_err(wg[c], wo[c]) < 1e-9 and rel_err(pg[c], po[c]) < 1e-9, c
    assert np.abs(hg - ho).max() < 1e-9 * np.abs(ho).max()
    assert np.abs(trg - tro).max() <= 1e-9 * max(np.abs(tro).max(), 1e-30) + 1e-18


def test_outlier_row_is_sent_to_fp64_by_the_certificate(hip, oracle):
    """One data row 1000x the others raises the pair exponents and coarsens the fixed-point grid of every other row: the certificate
    exceeds 1e-9, so a context created the way the shims create it (INT8_CERTIFY) runs this data on the fp64 kernels and matches the
    oracle to fp64 accuracy; a forced int8 context still satisfies the documented element-wise bound."""
    M, D, n = 6000, 20, 130
    XX, t = synthetic_logreg(M, D, 2)
    XX = XX.copy(); XX[1234] *= 1e3
    rs = np.random.RandomState(3)
    w = 0.3 * rs.randn(n, D) / np.sqrt(D); p = rs.randn(n, D)

    def fn(ctx):
        return ctx.metric(w) + ctx.leapfrog(w, p, 0.05, 1, 1, 4)

    auto = _capi.auto_metric_flags(D, 100000, M=M)                              # what RMHMC() / sample_sharded() pass for a big batch
    assert auto & _capi.FLAG_INT8_METRIC and auto & _capi.FLAG_INT8_CERTIFY
    (g, (bound, active)), (o, _) = _both(hip, oracle, XX, t, n, fn, auto)
    assert bound > _capi.INT8_CERTIFY_TOL and not active
    for a, b in zip(g, o):
        assert _nan_rel(a, b) < 1e-11
    (g8, (bound8, active8)), _ = _both(hip, oracle, XX, t, n, fn, _capi.int8_metric_flags(S))
    assert active8 and bound8 == bound
    B = _bound_matrix(XX, S)
    assert (np.abs(g8[0] - o[0]) <= B[None]).all()
    # and ordinary data of the same shape IS certified
    XX2, t2 = synthetic_logreg(M, D, 2)
    (_, (b2, a2)), _ = _both(hip, oracle, XX2, t2, n, lambda ctx: ctx.metric(w), auto)
    assert a2 and b2 < 1e-10


def test_two_contexts_of_different_shapes_stay_usable(hip, oracle):
    """the dynamic-LDS limit is per (function, device), not per context.  Two live contexts of the same kernel family with
    different LDS needs (k_step_medium: 8 < D <= 32, small batch) must both keep working whichever was created last."""
    out = {}
    shapes = [(2040, 16, 3), (130, 16, 3)]
    ctxs = []
    for M, D, n in shapes:
        XX, t = synthetic_logreg(M, D, 7)
        c = hip.context(M, D, n, flags=0); c.set_data(XX, t); ctxs.append((c, XX, t, M, D, n))
    rs = np.random.RandomState(0)
    for c, XX, t, M, D, n in ctxs + ctxs[::-1]:
        w = 0.1 * rs.randn(n, D); p = rs.randn(n, D)
        got = c.leapfrog(w, p, 0.4, 1, 2, 4)
        with oracle.context(M, D, n, flags=0) as oc:
            oc.set_data(XX, t)
            ref = oc.leapfrog(w, p, 0.4, 1, 2, 4)
        assert rel_err(got[0], ref[0]) < 1e-9 and rel_err(got[1], ref[1]) < 1e-9
        st = c.sample(12, 4, 3, 0.4, 4, seed=1)
        assert np.isfinite(st[0]).all()
    for c, *_ in ctxs:
        c.close()

"""int8 matrix-core metric assembly (RMHMC_FLAG_INT8_METRIC, csrc/metric_i8.hip.h) against the fp64 oracle and the
reference's golden vectors.  The assembly is an exact integer GEMM on S byte slices per operand, so its only error is the
fixed-point truncation 2^-(8S-2): the tests below hold S = 6 and 7 to the tolerances of the fp64 path (1e-9 after one leapfrog
step, where north_star asks for 1e-6) and S = 5 to 1e-9 as well; S = 4 (4e-10 on G) is only checked against 1e-6.
Needs an MI355X: run with  pytest -m gpu."""
import numpy as np
import pytest

from conftest import TAPES, load_tape, logdet_after_first_step, mat_err, rel_err
from riemannhamiltonianmontecarlo_amd import _capi
from riemannhamiltonianmontecarlo_amd.data import synthetic_logreg

pytestmark = pytest.mark.gpu

G_TOL = {4: 1e-8, 5: 1e-10, 6: 1e-12, 7: 1e-12}     # norm-wise, measured 2e-9 / 8e-12 / 3e-14 / 1e-14
STEP_TOL = {4: 1e-6, 5: 1e-9, 6: 1e-9, 7: 1e-9}     # theta, p, log|G| after one leapfrog step
SHAPES = [(1000, 64, 130), (690, 15, 5), (203, 33, 7), (129, 48, 300), (532, 12, 9), (50, 5, 64), (3000, 25, 257),
          (20, 10, 3), (33, 16, 130), (64, 17, 1), (97, 64, 129)]   # one / two k-stages, single chain, tile edges


def _run(lib, M, D, n, XX, t, fn, flags, options=None):
    with lib.context(M, D, n, flags=flags, options=options) as ctx:
        ctx.set_data(XX, t, 100.0)
        return fn(ctx)


@pytest.mark.parametrize("S", [4, 5, 6, 7])
@pytest.mark.parametrize("M,D,n", SHAPES)
def test_metric_and_leapfrog_match_oracle(hip, oracle, M, D, n, S):
    XX, t = synthetic_logreg(M, D, 1)
    rs = np.random.RandomState(M + D + S)
    w = 0.4 * rs.randn(n, D) / np.sqrt(D); p = rs.randn(n, D)
    dirs = np.where(rs.rand(n) < 0.5, -1, 1).astype(np.int32)

    def fn(ctx):
        return ctx.metric(w) + ctx.leapfrog(w, p, 0.5, dirs, 1, 4) + ctx.metric_terms(w, p)

    Gg, hg, gg, wg, pg, h1g, sg, trg, qg = _run(hip, M, D, n, XX, t, fn, _capi.int8_metric_flags(S))
    Go, ho, go, wo, po, h1o, so, tro, qo = _run(oracle, M, D, n, XX, t, fn, 0)
    # trace term tr(G^-1 dG_d): h_n = x_n' G^-1 x_n from the transposed sliced GEMM (error relative to max|G^-1| max|x_a x_b|)
    for c in range(n):
        assert rel_err(trg[c], tro[c]) < 1e3 * G_TOL[S], c
        assert rel_err(qg[c], qo[c]) < 1e3 * G_TOL[S], c
    assert np.array_equal(Gg, np.swapaxes(Gg, 1, 2))
    for c in range(n):
        assert rel_err(Gg[c], Go[c]) < G_TOL[S], c
        assert rel_err(wg[c], wo[c]) < STEP_TOL[S], c
        assert rel_err(pg[c], po[c]) < STEP_TOL[S], c
    assert np.abs(hg - ho).max() < 1e3 * G_TOL[S] and rel_err(gg, go) < 1e-12
    assert np.abs((h1g - h1o) / np.maximum(1.0, np.abs(h1o))).max() < STEP_TOL[S]


@pytest.mark.parametrize("S", [5, 6])
@pytest.mark.parametrize("name", TAPES)
def test_transitions_match_reference_golden(hip, name, S):
    """The reference's own draws and outputs (same checks as tests/test_gpu_parity.py for the fp64 assembly)."""
    XX, t, g = load_tape(name)
    T, D = g["z"].shape
    u_acc = np.where(np.isnan(g["u_acc"]), 0.5, g["u_acc"])
    with hip.context(XX.shape[0], D, T, flags=_capi.COMPAT | _capi.int8_metric_flags(S)) as ctx:
        ctx.set_data(XX, t, 100.0)
        r = ctx.transition(g["w_before"], g["z"], g["u_len"], g["g_dir"], u_acc, L=int(g["L"]), eps=float(g["eps"]), K=int(g["K"]))
    assert np.array_equal(r["nsteps"], g["nsteps"])
    finite = np.isfinite(g["H_prop"])
    for it in range(T):
        if not finite[it]:
            assert r["accepted"][it] == 0 and np.array_equal(r["w"][it], g["w_before"][it])
            continue
        e = max(rel_err(r["w_prop"][it], g["w_prop"][it]), rel_err(r["p_prop"][it], g["p_prop"][it]),
                abs(r["hld_prop"][it] - g["hld_prop"][it]) / max(1, abs(g["hld_prop"][it])))
        # whole trajectories (up to 6 steps); the australian and guard tapes pass through regions where G is nearly singular
        # and amplify the 8e-12 of 5 slices to 1e-7: 5 slices are held to the north_star bar, 6 to the fp64 path's tolerance
        tol = 1e-6 if S == 5 else 1e-8
        assert e < tol, (it, e)
        assert abs(r["H_prop"][it] - g["H_prop"][it]) < 10 * tol * max(1, abs(g["H_prop"][it])), it
        assert rel_err(r["w"][it], g["w_after"][it]) < tol, it


@pytest.mark.parametrize("name", ["australian", "syn_m203_d33", "syn_m10000_d64_L1", "syn_m3001_d130", "syn_m50000_d256_L1"])
def test_one_leapfrog_step_theta_and_logdet_vs_reference(hip, name):
    """north_star parity statement with the int8 assembly, 5 slices: theta and log|G| after ONE step vs the reference's values."""
    XX, t, g = load_tape(name)
    D = XX.shape[1]
    with hip.context(XX.shape[0], D, 1, flags=_capi.COMPAT | _capi.int8_metric_flags(5)) as ctx:
        ctx.set_data(XX, t, 100.0)
        w1, p1, hld1, st = ctx.leapfrog(g["w_before"][0], g["p0"][0], float(g["eps"]), int(g["dir"][0]), 1, int(g["K"]))
        G1, _, _ = ctx.metric(w1)
    assert rel_err(w1[0], g["it0_s0_w_end"]) < 1e-9 and rel_err(p1[0], g["it0_s0_p_end"]) < 1e-9
    assert mat_err(G1[0], g, "it0_s0_G_end") < 1e-9
    logdet_ref = logdet_after_first_step(g)
    assert abs(2 * hld1[0] - logdet_ref) < 1e-9 * max(1, abs(logdet_ref))


def test_nonfinite_chain_is_rejected_and_isolated(hip):
    """A NaN position makes v non-finite: that chain's G must come out NaN (rejected, flagged), its neighbours in the same
    128-chain tile must be bit-identical to a run without it."""
    M, D, n = 400, 20, 140
    XX, t = synthetic_logreg(M, D, 6)
    rs = np.random.RandomState(2)
    w = 0.05 * rs.randn(n, D); z = rs.randn(n, D)
    ul = rs.rand(n); gd = rs.randn(n); ua = rs.rand(n)
    wbad = w.copy(); wbad[17, 3] = np.nan
    fl = _capi.int8_metric_flags(6)
    good = _run(hip, M, D, n, XX, t, lambda c: c.transition(w, z, ul, gd, ua, L=3, eps=0.5, K=4), fl)
    bad = _run(hip, M, D, n, XX, t, lambda c: c.transition(wbad, z, ul, gd, ua, L=3, eps=0.5, K=4), fl)
    assert bad["accepted"][17] == 0 and bad["status"][17] != 0
    keep = np.arange(n) != 17
    for k in ("w", "w_prop", "H_prop", "accepted"):
        assert np.array_equal(bad[k][keep], good[k][keep]), k


def test_sampler_matches_oracle(hip, oracle):
    """Whole chains with shared Philox streams, int8 assembly with 6 slices vs the fp64 oracle."""
    M, D, n = 600, 16, 150
    XX, t = synthetic_logreg(M, D, 4)
    s1, a1, st1, _ = _run(hip, M, D, n, XX, t, lambda c: c.sample(25, 5, 4, 0.5, 4, seed=9), _capi.int8_metric_flags(6))
    s0, a0, st0, _ = _run(oracle, M, D, n, XX, t, lambda c: c.sample(25, 5, 4, 0.5, 4, seed=9), 0)
    assert np.array_equal(a1, a0) and np.array_equal(st1, st0)
    assert rel_err(s1, s0) < 1e-7


@pytest.mark.parametrize("M,D,n", [(300, 100, 4), (700, 256, 6), (500, 65, 130), (1200, 130, 33)])
def test_large_d_path(hip, oracle, M, D, n):
    """64 < D <= 256: the same sliced GEMMs feed the blocked Cholesky / inverse kernels of the large-D path."""
    XX, t = synthetic_logreg(M, D, 2)
    rs = np.random.RandomState(1)
    w = 0.1 * rs.randn(n, D) / np.sqrt(D); p = rs.randn(n, D)

    def fn(ctx):
        return ctx.metric(w) + ctx.metric_terms(w, p) + ctx.leapfrog(w, p, 0.3, 1, 1, 4)

    Gg, hg, gg, trg, qg, wg, pg, h1g, sg = _run(hip, M, D, n, XX, t, fn, _capi.int8_metric_flags(6))
    Go, ho, go, tro, qo, wo, po, h1o, so = _run(oracle, M, D, n, XX, t, fn, 0)
    for c in range(n):
        assert rel_err(Gg[c], Go[c]) < 1e-12, c
        assert rel_err(trg[c], tro[c]) < 1e-9, c
        assert rel_err(wg[c], wo[c]) < 1e-9 and rel_err(pg[c], po[c]) < 1e-9, c
    assert np.abs(hg - ho).max() < 1e-9


def test_long_contraction_is_split(hip, oracle):
    """M above the single-launch bound (21845 rows at 6 slices, 18724 at 7): the k range is summed over several launches."""
    M, D, n = 40000, 12, 130
    XX, t = synthetic_logreg(M, D, 5)
    rs = np.random.RandomState(4)
    w = 0.2 * rs.randn(n, D) / np.sqrt(D)
    Gg, hg, _ = _run(hip, M, D, n, XX, t, lambda c: c.metric(w), _capi.int8_metric_flags(7))
    Go, ho, _ = _run(oracle, M, D, n, XX, t, lambda c: c.metric(w), 0)
    for c in range(n):
        assert rel_err(Gg[c], Go[c]) < 1e-12, c


def test_full_size_config3(hip, oracle):
    """BASELINE config 3 size with the int8 assembly (6 slices): chains of a residue class agree bit for bit, eight distinct chains
    are checked against the oracle."""
    M, D, n, R = 10000, 64, 8192, 8
    XX, t = synthetic_logreg(M, D, 0)
    rs = np.random.RandomState(3)
    w8 = 0.05 * rs.randn(R, D); z8 = rs.randn(R, D); ul8 = rs.rand(R); gd8 = rs.randn(R); ua8 = rs.rand(R)
    rep = lambda a: np.ascontiguousarray(np.tile(a, (n // R,) + (1,) * (a.ndim - 1)))
    r = _run(hip, M, D, n, XX, t, lambda c: c.transition(rep(w8), rep(z8), rep(ul8), rep(gd8), rep(ua8), L=2, eps=0.5, K=4),
             _capi.int8_metric_flags(6))
    for k in ("w_prop", "p_prop", "H_prop", "w"):
        a = r[k].reshape((n // R, R) + r[k].shape[1:])
        assert np.array_equal(a, np.broadcast_to(a[0], a.shape)), k
    o = _run(oracle, M, D, R, XX, t, lambda c: c.transition(w8, z8, ul8, gd8, ua8, L=2, eps=0.5, K=4), 0)
    assert np.array_equal(r["nsteps"][:R], o["nsteps"]) and np.array_equal(r["accepted"][:R], o["accepted"])
    assert rel_err(r["w_prop"][:R], o["w_prop"]) < 1e-8 and rel_err(r["hld_prop"][:R], o["hld_prop"]) < 1e-8
    # proposal momentum, both Hamiltonians and the state after the accept step, like the fp64 twin (tests/test_gpu_parity.py)
    assert rel_err(r["p_prop"][:R], o["p_prop"]) < 1e-8 and rel_err(r["w"][:R], o["w"]) < 1e-8
    assert np.max(np.abs(r["H_prop"][:R] - o["H_prop"]) / np.maximum(1.0, np.abs(o["H_prop"]))) < 1e-8
    assert np.max(np.abs(r["H_cur"][:R] - o["H_cur"]) / np.maximum(1.0, np.abs(o["H_cur"]))) < 1e-9
    # element-wise on theta (rel_err is norm-wise): components above 1e-3 of the largest, each to 1e-7
    big = np.abs(o["w_prop"]) > 1e-3 * np.abs(o["w_prop"]).max()
    assert np.max(np.abs(r["w_prop"][:R][big] - o["w_prop"][big]) / np.abs(o["w_prop"][big])) < 1e-7


@pytest.mark.parametrize("M,D,n,S", [(400, 40, 300, 6), (1000, 40, 2432, 6), (400, 40, 300, 5), (900, 64, 2100, 6)])
def test_ragged_pair_block_as_tiles_of_its_own_is_bit_identical(hip, M, D, n, S):
    """The pairs beyond the last full block of 128 (D = 64: 2080 = 16 x 128 + 32) run as two-wave tiles of their own, k range in
    pieces summed as integers (k_assemble_i8_tail / _tailsum): same bits as the one-launch form, for one and for several pieces
    (2432 chains x 7 pair blocks and 2100 x 17: no k split of the main launch, 4 / 3 tail pieces)."""
    XX, t = synthetic_logreg(M, D, 3)
    rs = np.random.RandomState(n + S)
    w = 0.4 * rs.randn(n, D) / np.sqrt(D); p = rs.randn(n, D)
    dirs = np.where(rs.rand(n) < 0.5, -1, 1).astype(np.int32)
    out = {}
    for tail in ("0", "1"):
        with hip.context(M, D, n, flags=_capi.int8_metric_flags(S), options={"i8_tail": int(tail)}) as ctx:
            ctx.set_data(XX, t, 100.0)
            out[tail] = ctx.metric(w) + ctx.leapfrog(w, p, 0.5, dirs, 2, 4)
    for a, b in zip(out["0"], out["1"]):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("M,D,n", [(1000, 64, 130), (203, 33, 7), (3000, 25, 257), (10000, 64, 128)])
def test_inner_iterates_from_five_slices(hip, oracle, M, D, n):
    """At 6 slices the metric of the position fixed-point iterates before the last one is summed from the 5 most significant slices
    (it only steers the next iterate; RMHMC_FLAG_INT8_INNER_FULL turns that off).  G, log det and everything else of an EVALUATION
    point are untouched (bit-identical), theta / p after leapfrog steps move by < 1e-11, and both stay within 1e-9 of the oracle."""
    XX, t = synthetic_logreg(M, D, 5)
    rs = np.random.RandomState(M + n)
    w = 0.4 * rs.randn(n, D) / np.sqrt(D); p = rs.randn(n, D)
    dirs = np.where(rs.rand(n) < 0.5, -1, 1).astype(np.int32)

    def fn(ctx):
        return ctx.metric(w) + ctx.leapfrog(w, p, 0.5, dirs, 3, 4)

    fast = _run(hip, M, D, n, XX, t, fn, _capi.int8_metric_flags(6))
    full = _run(hip, M, D, n, XX, t, fn, _capi.int8_metric_flags(6) | _capi.FLAG_INT8_INNER_FULL)
    ref = _run(oracle, M, D, n, XX, t, fn, 0)
    for k in range(3):   # G, half log det, gradient at w
        assert np.array_equal(fast[k], full[k])
    assert not np.array_equal(fast[3], full[3])   # (the five-slice iterates are really in use)
    for c in range(n):
        for k in (3, 4):   # theta, p after three steps
            assert rel_err(fast[k][c], full[k][c]) < 1e-11, (c, k)
            assert rel_err(fast[k][c], ref[k][c]) < 1e-9 and rel_err(full[k][c], ref[k][c]) < 1e-9, (c, k)


# (shapes whose assembly is one launch over all data rows: with a k split - few tiles, many rows - the library keeps the full assembly)
# (not M = 97 < 2 D: its third step diverges to |theta| ~ 1e16 and multiplies ANY rounding difference by 1e5 - tools/diag_delta.py)
# (D > 64: the large-D path, whose slices are cut by k_vsplit and whose base matrix is a copy - Gq is factored in place)
@pytest.mark.parametrize("M,D,n", [(900, 64, 2100), (203, 33, 7), (400, 40, 2432), (129, 48, 300), (600, 96, 140), (900, 130, 260),
                                   (1200, 256, 130)])
def test_delta_assembly_at_the_end_of_a_step(hip, oracle, M, D, n):
    """The metric of the evaluation that ends a leapfrog step is G(last position iterate) plus the assembly of the DIFFERENCE of
    the two v vectors, cut into as few slices as its largest element needs (launch_assemble / I8Delta; option i8_delta = 0 turns it
    off).  Integer arithmetic on the same grids: theta / p / log det after three steps agree with the full assembly to the fp64
    rounding of one addition per step times the conditioning of the problem (`delta_end`: 3e-14; 3e-12 at M = 129 < 3 D; asserted
    1e-10), also when every chain is treated as re-based (its slices hold N itself and its G is overwritten: the path of a chain
    whose v exponent has changed).  With the second position iterate assembled as a delta of the first as well (`delta`: an inner
    iterate, five-slice accuracy like the full assembly of that iterate) the result moves like it does between five and six
    slices for the inner iterates (3e-13 at config 3's shape, 8e-11 at M = 129; asserted 5e-10).  All within 1e-9 of the oracle."""
    XX, t = synthetic_logreg(M, D, 5)
    rs = np.random.RandomState(M + n)
    w = 0.4 * rs.randn(n, D) / np.sqrt(D); p = rs.randn(n, D)
    dirs = np.where(rs.rand(n) < 0.5, -1, 1).astype(np.int32)

    def fn(ctx):
        return ctx.leapfrog(w, p, 0.5, dirs, 3, 4)

    out = {}
    variants = (("full", {"i8_delta": 0}, 0.0),
                ("delta_end", {"i8_delta": 1, "i8_delta_inner": 0}, 1e-10),
                ("rebase_end", {"i8_delta": 1, "i8_delta_inner": 0, "i8_force_rebase": 1}, 1e-10),
                ("delta", {"i8_delta": 1, "i8_delta_inner": 1}, 5e-10),
                ("rebase", {"i8_delta": 1, "i8_delta_inner": 1, "i8_force_rebase": 1}, 5e-10))
    for name, opts, _ in variants:
        out[name] = _run(hip, M, D, n, XX, t, fn, _capi.int8_metric_flags(6), options=opts)
    ref = _run(oracle, M, D, n, XX, t, fn, 0)
    assert any(not np.array_equal(a, b) for a, b in zip(out["full"], out["delta_end"]))   # (the delta paths are really in use)
    assert any(not np.array_equal(a, b) for a, b in zip(out["delta_end"], out["delta"]))
    for name, _, tol in variants[1:]:
        for k, (a, b, r) in enumerate(zip(out[name], out["full"], ref)):
            a, b, r = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64), np.asarray(r, dtype=np.float64)
            if a.ndim == 2:
                for c in range(n):
                    assert rel_err(a[c], b[c]) < tol, (name, k, c)
                    assert rel_err(a[c], r[c]) < 1e-9, (name, k, c)
            else:
                assert np.allclose(a, b, rtol=tol, atol=tol), (name, k)
                assert np.allclose(a, r, rtol=1e-9, atol=1e-9), (name, k)

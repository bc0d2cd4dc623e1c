"""BASELINE config 5 at its own size (4096 chains, D = 256, M = 50000) on the paths `bench.py --workload c5` uses: the int8 large-D
path with its multi-launch `accumulate` epilogue (M = 50000 > 21845 rows per overflow-safe launch: 3 assembly launches; 32896
column pairs: 2 leverage launches) and the fp64 blocked path.  The oracle runs a handful of chains (a D = 256, M = 50000 leapfrog
step is ~20 GF on the CPU); the other chains are covered by residue-class replication: chains fed the same inputs must agree
bit for bit wherever they sit in the batch.  Reference blocks: rmhmc.py:57,119,137 (assembly), :64-77 (leverage / trace term).
Needs an MI355X: run with  pytest -m gpu."""
import numpy as np
import pytest

from conftest import rel_err
from riemannhamiltonianmontecarlo_amd import _capi
from riemannhamiltonianmontecarlo_amd.data import synthetic_logreg

pytestmark = pytest.mark.gpu

M5, D5 = 50000, 256
_cache = {}


def _data():
    if "xt" not in _cache:
        _cache["xt"] = synthetic_logreg(M5, D5, 0)
    return _cache["xt"]


def test_config5_int8_long_contraction_matches_oracle(hip, oracle):
    """D = 256, M = 50000, 6 slices, 130 chains (two 128-chain tiles, the second nearly empty): G, trace term, theta / p / log|G| after
    one leapfrog step against the fp64 oracle at 1e-9 on chains at the tile edges and inside."""
    XX, t = _data()
    n = 130
    pick = np.array([0, 1, 63, 127, 128, 129])
    rs = np.random.RandomState(5)
    w = 0.1 * rs.randn(n, D5) / np.sqrt(D5); p = rs.randn(n, D5)
    dirs = np.where(rs.rand(n) < 0.5, -1, 1).astype(np.int32)
    with hip.context(M5, D5, n, flags=_capi.int8_metric_flags(6)) as ctx:
        ctx.set_data(XX, t)
        assert ctx.int8_certificate()[1]
        Gg, hg, gg = ctx.metric(w)
        trg, qg = ctx.metric_terms(w, p)
        wg, pg, h1g, sg = ctx.leapfrog(w, p, 0.3, dirs, 1, 4)
    with oracle.context(M5, D5, len(pick), flags=0) as ctx:
        ctx.set_data(XX, t)
        Go, ho, go = ctx.metric(w[pick])
        tro, qo = ctx.metric_terms(w[pick], p[pick])
        wo, po, h1o, so = ctx.leapfrog(w[pick], p[pick], 0.3, dirs[pick], 1, 4)
    assert np.array_equal(Gg, np.swapaxes(Gg, 1, 2)) and np.isfinite(wg).all() and not sg.any()
    for k, c in enumerate(pick):
        assert rel_err(Gg[c], Go[k]) < 1e-12, c
        assert rel_err(gg[c], go[k]) < 1e-11, c
        assert rel_err(trg[c], tro[k]) < 1e-9 and rel_err(qg[c], qo[k]) < 1e-9, c
        assert rel_err(wg[c], wo[k]) < 1e-9 and rel_err(pg[c], po[k]) < 1e-9, c
    assert np.abs(hg[pick] - ho).max() < 1e-9 * np.abs(ho).max()
    assert np.abs(h1g[pick] - h1o).max() < 1e-9 * np.abs(h1o).max()


def _full_size_inputs():
    R = 8
    rs = np.random.RandomState(17)
    return R, dict(w=0.02 * rs.randn(R, D5), z=rs.randn(R, D5), ul=rs.rand(R), gd=rs.randn(R), ua=rs.rand(R))


def _oracle_full_size(oracle):
    if "o5" not in _cache:
        XX, t = _data()
        R, i = _full_size_inputs()
        with oracle.context(M5, D5, R, flags=0) as ctx:
            ctx.set_data(XX, t)
            _cache["o5"] = ctx.transition(i["w"], i["z"], i["ul"], i["gd"], i["ua"], L=2, eps=0.5, K=4)
    return _cache["o5"]


@pytest.mark.parametrize("slices", [0, 6])
def test_config5_full_size(hip, oracle, slices):
    """The whole config: 4096 chains x D 256 x M 50000, one transition of up to two leapfrog steps, fp64 blocked path (slices = 0) and
    int8 path (6 slices).  Chains of a residue class mod 8 get the same inputs and must agree bit for bit; one class representative
    each is checked against the oracle (theta, p, log|G| of the proposal, Hamiltonian, accept decision)."""
    XX, t = _data()
    n = 4096
    R, i = _full_size_inputs()
    rep = lambda a: np.ascontiguousarray(np.tile(a, (n // R,) + (1,) * (a.ndim - 1)))
    flags = _capi.int8_metric_flags(slices) if slices else 0
    with hip.context(M5, D5, n, flags=flags) as ctx:
        ctx.set_data(XX, t)
        r = ctx.transition(rep(i["w"]), rep(i["z"]), rep(i["ul"]), rep(i["gd"]), rep(i["ua"]), L=2, eps=0.5, K=4)
    for k in ("w_prop", "p_prop", "H_prop", "hld_prop", "w", "accepted", "nsteps"):
        a = r[k].reshape((n // R, R) + r[k].shape[1:])
        assert np.array_equal(a, np.broadcast_to(a[0], a.shape)), k
    o = _oracle_full_size(oracle)
    assert np.array_equal(r["nsteps"][:R], o["nsteps"]) and np.array_equal(r["accepted"][:R], o["accepted"])
    assert set(o["nsteps"]) == {1, 2}          # both trajectory lengths occur
    assert rel_err(r["w_prop"][:R], o["w_prop"]) < 1e-9 and rel_err(r["p_prop"][:R], o["p_prop"]) < 1e-9
    assert np.abs(r["hld_prop"][:R] - o["hld_prop"]).max() < 1e-9 * np.abs(o["hld_prop"]).max()
    assert np.abs(r["H_prop"][:R] - o["H_prop"]).max() < 1e-9 * np.abs(o["H_prop"]).max()
    assert rel_err(r["w"][:R], o["w"]) < 1e-9

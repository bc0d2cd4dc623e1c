"""ESS / autocorrelation restatement vs outputs of the reference's tools.py (golden)."""
import os

import numpy as np

from conftest import GOLDEN
from riemannhamiltonianmontecarlo_amd import tools


def test_ac_and_ess_match_reference_ar1():
    g = np.load(os.path.join(GOLDEN, "ess_ar1.npz"))
    x, lag = g["samples"], int(g["maxlag"])
    acf = np.stack([tools.ac(x[:, j], lag) for j in range(x.shape[1])], axis=1)
    assert np.allclose(acf, g["acf"], rtol=1e-10, atol=1e-12)
    ess = tools.CalculateESS(x, lag)
    assert ess.shape == (x.shape[1], 1)
    assert np.allclose(ess.ravel(), g["ess"], rtol=1e-9)
    # sanity of the estimator itself: more autocorrelation -> fewer effective samples
    assert ess[0] > ess[1] > ess[2]


def test_ess_matches_reference_on_a_reference_chain():
    g = np.load(os.path.join(GOLDEN, "ess_pima_chain.npz"))
    ess = tools.CalculateESS(g["samples"], int(g["maxlag"]))
    assert np.allclose(ess.ravel(), g["ess"], rtol=1e-9)


def test_matlab_fft_length_has_no_wraparound():
    rs = np.random.RandomState(0)
    x = rs.randn(500, 2)
    a = tools.CalculateESS(x, 499, nfft="python")
    b = tools.CalculateESS(x, 499, nfft="matlab")
    assert a.shape == b.shape == (2, 1) and np.all(b > 100)
    assert tools.min_ess_per_chain(x[None], nfft="matlab").shape == (1,)


def test_lognormpdf():
    w = np.array([[0.1], [-0.2], [0.3]])
    v = tools.LogNormPDF(np.zeros((1, 3)), w, 100.0)
    assert abs(v - np.sum(-0.5 * np.log(2 * np.pi * 100) - w ** 2 / 200)) < 1e-14


def test_oracle_direct_ess_matches_reference(oracle):
    """rmhmc_ess / rmhmc_sample_stats of the C-ABI (direct lag-by-lag evaluation, no FFT) against the outputs of the
    reference's tools.CalculateESS (golden) and against the host restatement."""
    rs = np.random.RandomState(1)
    with oracle.context(10, 2, 1) as ctx:
        for name in ("ess_ar1", "ess_pima_chain"):
            g = np.load(os.path.join(GOLDEN, name + ".npz"))
            assert np.allclose(ctx.ess(g["samples"])[0], g["ess"], rtol=1e-10)
        x = np.cumsum(rs.randn(3, 401, 4), axis=1) * 0.1 + rs.randn(3, 401, 4)   # odd S
        ref = np.stack([tools.CalculateESS(x[i], 400, nfft="matlab").ravel() for i in range(3)])
        assert np.allclose(ctx.ess(x), ref, rtol=1e-10)
    d = np.load(os.path.join(GOLDEN, "data_pima.npz"))
    with oracle.context(d["XX"].shape[0], d["XX"].shape[1], 3) as ctx:
        ctx.set_data(d["XX"], d["t"])
        s, acc, steps, _ = ctx.sample(40, 10, seed=2)
        st = ctx.sample_stats(40, 10, seed=2)
    assert np.allclose(st["mean"], s.mean(1)) and np.allclose(st["var"], s.var(1))
    assert np.array_equal(st["accepted"], acc) and np.array_equal(st["leapfrog_steps"], steps)
    ref = np.stack([tools.CalculateESS(s[i], 29, nfft="matlab").ravel() for i in range(3)])
    assert np.allclose(st["ess"], ref, rtol=1e-9)


def test_python_fft_length_wraparound_is_reproduced(oracle):
    """S = 4096, a power of two: the reference's nFFT = nextpow2(S)+1 = 4097 (tools.py:16-23) wraps lag nFFT-l onto lag l.  The host
    restatement (nfft="python") and the C-ABI's direct evaluation with RMHMC_FLAG_ESS_WRAP reproduce the reference's outputs; the
    default (linear autocovariances, the MATLAB original) is a different, documented estimator."""
    g = np.load(os.path.join(GOLDEN, "ess_s4096.npz"))
    x = g["samples"]
    acf = np.stack([tools.ac(x[:, j], 64) for j in range(x.shape[1])], axis=1)
    assert np.allclose(acf, g["acf64"], rtol=1e-10, atol=1e-13)
    assert np.allclose(tools.CalculateESS(x, int(g["maxlag"])).ravel(), g["ess"], rtol=1e-9)
    from riemannhamiltonianmontecarlo_amd import _capi
    with oracle.context(10, 2, 1, flags=_capi.FLAG_ESS_WRAP) as ctx:
        for name in ("ess_s4096", "ess_ar1", "ess_pima_chain"):
            gg = np.load(os.path.join(GOLDEN, name + ".npz"))
            assert np.allclose(ctx.ess(gg["samples"])[0], gg["ess"], rtol=1e-10), name
    # a short series where the two FFT lengths visibly disagree: S = 9 -> nFFT = 17, lag l collects lag 17 - l
    rs = np.random.RandomState(5)
    y = np.cumsum(rs.randn(9, 3), axis=0)
    with oracle.context(10, 2, 1, flags=_capi.FLAG_ESS_WRAP) as ctx:
        wrapped = ctx.ess(y)[0]
    with oracle.context(10, 2, 1, flags=0) as ctx:
        linear = ctx.ess(y)[0]
    assert np.allclose(wrapped, tools.CalculateESS(y, 8, nfft="python").ravel(), rtol=1e-10)
    assert np.allclose(linear, tools.CalculateESS(y, 8, nfft="matlab").ravel(), rtol=1e-10)

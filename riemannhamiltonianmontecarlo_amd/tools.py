"""ESS / autocorrelation: the min-ESS half of the benchmark metric.

Host-side restatement of the reference's code/tools.py:16-74 (itself a
translation of authors_code/Bayes_Log_Reg/Results/{ac,CalculateESS}.m),
vectorised over parameters.  ``nfft="python"`` reproduces the reference's FFT
length ``nextpow2(S)+1`` (tools.py:23 — an odd length that wraps lags beyond
nFFT-S); ``nfft="matlab"`` uses 2^(nextpow2+1) as ac.m:78 does (no wrap).
"""
import numpy as np


def nextpow2(i):
    """Smallest power of two >= i, returned as the power itself (tools.py:16-19)."""
    n = 1
    while n < i:
        n *= 2
    return n


def LogNormPDF(Values, Means, Variance):
    """Isotropic Gaussian log density summed over dimensions (tools.py:10-14)."""
    Values = np.asarray(Values, dtype=np.float64)
    if Values.ndim == 2 and Values.shape[1] > 1:
        Values = Values.T
    D = Values.shape[0]
    return float(np.sum(-0.5 * np.log(2 * np.pi * Variance) * np.ones((D, 1))
                        - (Values - Means) ** 2 / (2.0 * Variance)))


def ac(Series, nLag, nfft="python"):
    """Normalised autocorrelation for lags 0..nLag via FFT (tools.py:21-30)."""
    x = np.asarray(Series, dtype=np.float64).ravel()
    n = nextpow2(len(x)) + 1 if nfft == "python" else 2 * nextpow2(len(x))
    F = np.fft.fft(x - x.mean(), n)
    acf = np.real(np.fft.ifft(F * np.conj(F)))[: nLag + 1]
    return acf / acf[0]


def CalculateESS(Samples, MaxLag, nfft="python"):
    """Geyer initial-monotone-sequence ESS per column (tools.py:32-74).  Returns (P,1)."""
    Samples = np.asarray(Samples, dtype=np.float64)
    MaxLag = int(MaxLag)
    S, P = Samples.shape
    ACs = np.stack([ac(Samples[:, i], MaxLag, nfft) for i in range(P)], axis=1)
    half = (MaxLag + 1) // 2
    Gamma = ACs[0:2 * half:2] + ACs[1:2 * half:2]          # Gamma_j = rho_2j + rho_2j+1
    Gamma = np.minimum.accumulate(Gamma, axis=0)            # initial monotone sequence
    npos = (Gamma > 0).sum(axis=0)                          # tools.py:65-67 sums the FIRST npos entries
    csum = np.vstack([np.zeros((1, P)), np.cumsum(Gamma, axis=0)])
    mono = -ACs[0] + 2.0 * csum[npos, np.arange(P)]
    mono = np.where(mono < 1, 1.0, mono)
    return (S / mono).reshape(P, 1)


def min_ess_per_chain(samples, nfft="matlab"):
    """samples (n_chains, S, D) -> (n_chains,) min over dimensions of the per-chain ESS
    (MATLAB Results/CalculateStatistics.m:11-17 semantics: ESS per run, not of the run-mean)."""
    samples = np.asarray(samples)
    out = np.empty(samples.shape[0])
    for c in range(samples.shape[0]):
        out[c] = CalculateESS(samples[c], samples.shape[1] - 1, nfft).min()
    return out

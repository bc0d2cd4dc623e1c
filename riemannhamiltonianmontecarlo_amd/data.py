"""Input recipes for the RMHMC hot path.

``load_csv_dataset`` is the counterpart of the preprocessing block of the
reference driver (code/main.py:20-41): last CSV column is the label, labels
{1,2} are remapped to {0,1} (main.py:24-27; BLR_RMHMC.m:52-55 does the same for
german), covariates are z-scored (main.py:37) and an intercept column is
prepended (main.py:40-41).

``synthetic_logreg`` is the seeded synthetic recipe of SURVEY.md §8(d) used by
BASELINE configs 2-5 (no intercept, unit-variance columns).
"""
import numpy as np


def load_csv_dataset(path, polynomial_order=1):
    """polynomial_order=3 builds the cubic basis [1, X, X^2, X^3] the authors use for Ripley (D = 7,
    authors_code/Bayes_Log_Reg/MCMC/BLR_RMHMC.m:151-173); 1 is main.py's [1, X]."""
    raw = np.loadtxt(path, delimiter=",")
    t = raw[:, -1:].astype(np.float64).copy()
    if set(np.unique(t)) == {1.0, 2.0}:
        t = t - 1.0
    X = raw[:, :-1]
    X = (X - X.mean(axis=0)) / X.std(axis=0)
    XX = np.hstack([np.ones((X.shape[0], 1))] + [X ** i for i in range(1, polynomial_order + 1)])
    return np.ascontiguousarray(XX, dtype=np.float64), np.ascontiguousarray(t)


def synthetic_logreg(M, D, seed):
    """X ~ N(0,1)^(M x D), w* ~ N(0, I/D), t ~ Bernoulli(sigmoid(X w*)).  float64."""
    rs = np.random.RandomState(seed)
    X = rs.randn(M, D)
    w_true = rs.randn(D, 1) / np.sqrt(D)
    t = (rs.rand(M, 1) < 1.0 / (1.0 + np.exp(-X.dot(w_true)))).astype(np.float64)
    return np.ascontiguousarray(X), t

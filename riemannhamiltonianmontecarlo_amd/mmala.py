"""Simplified manifold MALA for the same model, the sampler the paper compares RMHMC with.

The repository holds it only as MATLAB, authors_code/Bayes_Log_Reg/MCMC/BLR_mMALA_Simp.m (:175-290; simplified=False: the
full sampler BLR_mMALA.m, whose drift also carries the derivative of the metric): w0 = 0,
StepSize 1, 10000 iterations of which 5000 burn-in, proposal N(w + eps/2 G^-1 grad, eps G^-1), Metropolis-
Hastings with the reverse proposal evaluated under G(w').  The Python code/ directory has no counterpart, so
this follows the naming and return convention of the two Python samplers:

    wSaved, TimeTaken = mMALA(XX, t, NumOfIterations=10000, BurnIn=5000, StepSize=1.0)

Parity is UNPINNED: no reference implementation is runnable here (MATLAB), so the GPU path is checked against
oracle/rmhmc_oracle.c's restatement, and the restatement against the paper's published ESS.
Runs on the MI355X through rmhmc_mmala_sample (include/rmhmc.h).  No CPU fallback.
"""
import numpy as np

from . import _capi


def mMALA(XX, t, NumOfIterations=10000, BurnIn=5000, StepSize=1.0, *, simplified=True, n_chains=1, seed=None, theta0=None,
          alpha=100.0, device=0, chain_offset=0, verbose=True, return_info=False, _lib=None):
    """ SIMPLIFIED MANIFOLD MALA (Bayesian logistic regression, N(0, alpha I) prior) """
    XX = np.ascontiguousarray(XX, dtype=np.float64)
    if XX.ndim != 2:
        raise ValueError("XX must be (N, D)")
    N, D = XX.shape
    t = np.ascontiguousarray(t, dtype=np.float64).reshape(-1)
    if t.shape[0] != N:
        raise ValueError("t must have N entries")
    if not BurnIn < NumOfIterations:
        raise ValueError("BurnIn must be smaller than NumOfIterations")
    if seed is None:
        seed = int(np.random.randint(0, 2 ** 62))
    lib = _lib if _lib is not None else _capi.load_hip_library()
    # simplified=False: the full manifold MALA of BLR_mMALA.m (:198-233): the drift gains the metric-derivative terms, which
    # collapse to eps/2 G^-1 (grad - trace term)
    with lib.context(N, D, n_chains, flags=0 if simplified else _capi.FLAG_MMALA_FULL, device=device) as ctx:
        ctx.set_data(XX, t, alpha)
        samples, acc, seconds = ctx.mmala_sample(NumOfIterations, BurnIn, StepSize, seed=seed, chain_offset=chain_offset,
                                                 theta0=theta0)
    if verbose:
        print('Acceptance: {}'.format(float(acc.sum()) / (NumOfIterations * n_chains)))
        print('Time drawing posterior: {}'.format(seconds))
    wSaved = samples[0] if n_chains == 1 else samples
    if return_info:
        return wSaved, seconds, dict(accepted=acc, seed=seed)
    return wSaved, seconds

"""Drop-in counterpart of the reference's plain HMC sampler, code/hmc.py (the sampler the
unchanged code/main.py actually calls at main.py:53).

    wSaved, TimeTaken = HMC(XX, t, NumOfIterations=6000, BurnIn=1000, NumOfLeapFrogSteps=100, StepSize=0.14)

Identity mass, theta0 = 0 (hmc.py:21,27), trajectory length ceil(rand*NumOfLeapFrogSteps) (hmc.py:48).  Runs on
the MI355X through rmhmc_hmc_sample (include/rmhmc.h); same keyword-only extensions and the same row-0
convention as riemannhamiltonianmontecarlo_amd.rmhmc.RMHMC.  No CPU fallback.
"""
import numpy as np

from . import _capi


def HMC(XX, t, NumOfIterations=6000, BurnIn=1000, NumOfLeapFrogSteps=100, StepSize=0.14, *, n_chains=1, seed=None,
        theta0=None, alpha=100.0, device=0, chain_offset=0, verbose=True, return_info=False, options=None, _lib=None):
    """ HAMILTONIAN MONTE CARLO (Bayesian logistic regression, N(0, alpha I) prior) """
    XX = np.ascontiguousarray(XX, dtype=np.float64)
    if XX.ndim != 2:
        raise ValueError("XX must be (N, D)")
    N, D = XX.shape
    t = np.ascontiguousarray(t, dtype=np.float64).reshape(-1)
    if t.shape[0] != N:
        raise ValueError("t must have N entries")
    if not BurnIn < NumOfIterations:
        raise ValueError("BurnIn must be smaller than NumOfIterations")  # NameError in the reference (hmc.py:92-96)
    if seed is None:
        seed = int(np.random.randint(0, 2 ** 62))
    lib = _lib if _lib is not None else _capi.load_hip_library()
    with lib.context(N, D, n_chains, flags=0, device=device, options=options) as ctx:
        ctx.set_data(XX, t, alpha)
        if verbose:  # the reference's stdout, hmc.py:85-89,92-94
            from .rmhmc import progress_printer
            ctx.set_progress(progress_printer(n_chains, hmc_burn_in=BurnIn), first=1, every=50)
        samples, acc, steps, seconds = ctx.hmc_sample(NumOfIterations, BurnIn, NumOfLeapFrogSteps, StepSize, seed=seed,
                                                      chain_offset=chain_offset, theta0=theta0)
    if verbose:
        print('Time drawing posterior: {}'.format(seconds))
    wSaved = samples[0] if n_chains == 1 else samples
    if return_info:
        return wSaved, seconds, dict(accepted=acc, leapfrog_steps=steps, seed=seed)
    return wSaved, seconds

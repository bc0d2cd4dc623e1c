"""Drop-in counterpart of the reference sampler module code/rmhmc.py.

``RMHMC`` keeps the reference signature and return contract (rmhmc.py:13,201):

    wSaved, TimeTaken = RMHMC(XX, t, NumOfIterations=6000, BurnIn=1000,
                              NumOfLeapFrogSteps=6, StepSize=0.5, NumOfNewtonSteps=4)

and runs every transition on the MI355X through the C-ABI of include/rmhmc.h
(librmhmc_hip.so).  There is no CPU fallback: without the built HIP library or
without a GPU the call raises.

Keyword-only extensions (defaults preserve the one-chain contract):
  n_chains   independent chains; n_chains>1 returns wSaved of shape (n_chains, S, D)
  seed       Philox key; None draws one from the global ``np.random`` stream, so
             ``np.random.seed(k)`` makes a run reproducible as it does for the reference
  compat     True reproduces the reference's p = L'z momentum and its two RENORMALIZE
             guards (rmhmc.py:80-85,125-130); False uses p = L z and no guards
  theta0     initial position(s); default 1e-3 everywhere (rmhmc.py:27)
  alpha      prior variance (rmhmc.py:19 hard-codes 100)
  device     HIP device ordinal
  int8_slices  metric assembly on the int8 matrix cores (include/rmhmc.h, RMHMC_FLAG_INT8_METRIC): 4..7 byte slices per
             operand, 0 = fp64 matrix cores, None (default) = 6 slices (G to 2e-14, the level of fp64 summation) when the
             path applies (8 < D <= 256) and there is enough work for its tiles (n_chains * N * D^2 >= 1e9), else fp64
  options    dict of the library's tuning options (include/rmhmc.h, rmhmc_create_opts), e.g. {"graph": 0}
  return_info  also return a dict(accepted=..., leapfrog_steps=...)

Documented deviations: row 0 of wSaved is undefined in the reference (np.empty,
never written: rmhmc.py:28,190-191); here it holds the state after iteration
``BurnIn``.  Random streams differ from NumPy's MT19937, so runs agree with the
reference in distribution, not sample by sample (single transitions with the
reference's own draws are checked bit-closely through ``rmhmc_transition``).
"""
import numpy as np

from . import _capi


def progress_printer(n_chains, hmc_burn_in=None):
    """The reference's stdout, driven by rmhmc_set_progress.  rmhmc.py:38-45,195: at the top of every iteration whose number+1 is a
    multiple of 50 - i.e. after 49, 99, ... completed transitions - the text '<number+1> iterations completed.' and the acceptance
    rate of the window since the last report (the first window holds 49 proposals), plus the burn-in banner.  hmc.py:85-89,92-94
    (hmc_burn_in given): after iterations 0, 50, 100, ... up to BurnIn, '<number> iterations completed.' and the window's rate."""
    last = {"itot": 0, "acc": 0}

    def report(event, iters, accepted, iters_total):
        if event == _capi.EV_BURNIN_DONE:
            print('Burn-in complete, now drawing posterior samples.')
            return
        if hmc_burn_in is not None and iters - 1 > hmc_burn_in:
            return
        print('{} iterations completed.'.format(iters + 1 if hmc_burn_in is None else iters - 1))
        # accepted / proposed since the last report, over all chains (one chain: exactly the reference's window)
        print('Acceptance: {}'.format((accepted - last["acc"]) / float(max(1, iters_total - last["itot"]))))
        last["itot"], last["acc"] = iters_total, accepted
    return report


def RMHMC(XX, t, NumOfIterations=6000, BurnIn=1000, NumOfLeapFrogSteps=6, StepSize=0.5, NumOfNewtonSteps=4, *,
          n_chains=1, seed=None, compat=True, theta0=None, alpha=100.0, device=0, chain_offset=0, verbose=True,
          return_info=False, int8_slices=None, options=None, _lib=None):
    """ RIEMANNIAN HAMILTONIAN MONTE CARLO (Bayesian logistic regression, N(0, alpha I) prior) """
    XX = np.ascontiguousarray(XX, dtype=np.float64)
    if XX.ndim != 2:
        raise ValueError("XX must be (N, D)")
    N, D = XX.shape
    t = np.ascontiguousarray(t, dtype=np.float64).reshape(-1)
    if t.shape[0] != N:
        raise ValueError("t must have N entries")
    if not BurnIn < NumOfIterations:
        # the reference raises NameError here (`start` unbound, rmhmc.py:194-198)
        raise ValueError("BurnIn must be smaller than NumOfIterations")
    if seed is None:
        seed = int(np.random.randint(0, 2 ** 62))
    lib = _lib if _lib is not None else _capi.load_hip_library()
    flags = (_capi.COMPAT if compat else 0) | _capi.auto_metric_flags(D, n_chains, int8_slices, M=N)
    with lib.context(N, D, n_chains, flags=flags, device=device, options=options) as ctx:
        ctx.set_data(XX, t, alpha)
        if verbose:
            ctx.set_progress(progress_printer(n_chains))
        samples, acc, steps, seconds = ctx.sample(NumOfIterations, BurnIn, NumOfLeapFrogSteps, StepSize,
                                                  NumOfNewtonSteps, seed=seed, chain_offset=chain_offset,
                                                  theta0=theta0)
    if verbose:
        print('Time drawing posterior: {}'.format(seconds))
    wSaved = samples[0] if n_chains == 1 else samples
    if return_info:
        return wSaved, seconds, dict(accepted=acc, leapfrog_steps=steps, seed=seed)
    return wSaved, seconds

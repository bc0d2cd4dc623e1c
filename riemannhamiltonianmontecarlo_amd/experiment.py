"""Counterpart of the reference's experiment driver, code/main.py:43-79.

The reference script runs one sampler ``n_experiments = 10`` times on the preprocessed data (main.py:43-53), averages the chains
and the times over the runs (:54-55), and reports the ESS of the RUN-MEAN chain with ``CalculateESS(avg, S-1)`` (:70-71), its
min / median / mean / max (:73-76), the mean time (:77) and "Time per Min ESS" (:79).  The interactive parts of the script
(``pdb.set_trace()`` :57, the matplotlib plots :62-67) are not reproduced.

``run_experiment`` returns the same quantities under the reference's own names, plus the per-run ESS that the MATLAB original
computes (``Results/CalculateStatistics.m:11-17``: ESS per run, then averaged) - the semantics ``bench.py`` uses for min-ESS/sec.

Two ways to execute the runs:
  batched=False  ``n_experiments`` sampler calls one after another, as main.py does: ``results_time[i]`` is run i's own TimeTaken;
  batched=True   all runs as independent chains of ONE sampler call on the GPU (chain i = run i; its Philox stream is keyed by the
                 chain id, so run i's samples are the same either way when ``seed`` is given): ``results_time[i]`` is the wall
                 time of the whole batch for every i - the runs finish together.
"""
import numpy as np

from . import tools
from .hmc import HMC
from .mmala import mMALA
from .rmhmc import RMHMC

SAMPLERS = {"RMHMC": RMHMC, "HMC": HMC, "mMALA": mMALA}


def summarize(results_beta, results_time, nfft="python"):
    """main.py:54-79 on given (results_beta (n_runs, S, D), results_time (n_runs,)).  ``nfft`` as in tools.CalculateESS: "python" is the
    reference's FFT length (tools.py:23)."""
    results_beta = np.asarray(results_beta, dtype=np.float64)
    results_time = np.asarray(results_time, dtype=np.float64)
    avg_beta_posterior = np.mean(results_beta, axis=0)                                   # main.py:54
    avg_time_taken = float(np.mean(results_time))                                        # main.py:55
    S = avg_beta_posterior.shape[0]
    ESS = tools.CalculateESS(avg_beta_posterior, S - 1, nfft)                            # main.py:71
    per_run = np.stack([tools.CalculateESS(results_beta[i], S - 1, nfft).ravel() for i in range(results_beta.shape[0])])
    return {
        "avg_beta_posterior": avg_beta_posterior, "avg_time_taken": avg_time_taken, "ESS": ESS,
        "Min": float(np.min(ESS)), "Median": float(np.median(ESS)), "Mean": float(np.mean(ESS)), "Max": float(np.max(ESS)),
        "Time": avg_time_taken, "Time per Min ESS": round(avg_time_taken / float(np.min(ESS)), 6),     # main.py:73-79
        # MATLAB semantics (CalculateStatistics.m:11-17): ESS of every run, statistics averaged over the runs
        "ESS_per_run": per_run,
        "per_run": {"Min": float(per_run.min(1).mean()), "Median": float(np.median(per_run, 1).mean()),
                    "Mean": float(per_run.mean(1).mean()), "Max": float(per_run.max(1).mean()),
                    "Time per Min ESS": float(avg_time_taken / per_run.min(1).mean())},
    }


def run_experiment(XX, t, sampler="RMHMC", n_experiments=10, batched=False, seed=None, verbose=False, nfft="python", **sampler_kwargs):
    """main.py:43-79.  Returns a dict with ``results_beta`` (n_experiments, S, D), ``results_time`` (n_experiments,) and the summary of
    ``summarize``.  ``sampler_kwargs`` go to the sampler (NumOfIterations, BurnIn, StepSize, compat, device, ...); the defaults are the
    samplers' own, i.e. S = 5000 rows per run as main.py:46 hard-codes."""
    fn = SAMPLERS[sampler] if isinstance(sampler, str) else sampler
    if seed is None:
        seed = int(np.random.randint(0, 2 ** 62))
    if batched:
        smp, secs = fn(XX, t, n_chains=n_experiments, seed=seed, verbose=verbose, **sampler_kwargs)
        results_beta = smp if n_experiments > 1 else smp[None]
        results_time = np.full(n_experiments, secs)
    else:
        runs, times = [], []
        for i in range(n_experiments):
            # run i = global chain i of the same seed: identical to chain i of the batched call
            smp, secs = fn(XX, t, n_chains=1, seed=seed, chain_offset=i, verbose=verbose, **sampler_kwargs)
            runs.append(smp); times.append(secs)
        results_beta = np.stack(runs); results_time = np.asarray(times)
    out = {"results_beta": results_beta, "results_time": results_time, "seed": seed, "sampler": getattr(fn, "__name__", str(fn))}
    out.update(summarize(results_beta, results_time, nfft))
    return out


def report(res, file=None):
    """The reference's print-out (main.py:70-79)."""
    print('ESS', file=file)
    for k in ('Min', 'Median', 'Mean', 'Max', 'Time'):
        print(k, res[k], file=file)
    print('Time per Min ESS:', res['Time per Min ESS'], file=file)

"""experiment.py, the counterpart of the reference driver's run / average / ESS block (code/main.py:43-79), on the CPU through the
oracle (injected as `_lib`; the product default is the HIP library)."""
import io
import os

import numpy as np

from conftest import GOLDEN
from riemannhamiltonianmontecarlo_amd import experiment, tools


def test_summary_follows_main_py():
    """main.py:54-79 on fixed inputs: ESS of the run-MEAN chain by tools.CalculateESS (pinned to the reference's outputs in
    test_tools_ess.py), its min / median / mean / max, mean time, rounded time per min ESS; plus the per-run (MATLAB) statistics."""
    g = np.load(os.path.join(GOLDEN, "ess_pima_chain.npz"))
    x = g["samples"]                                   # a chain produced by the reference itself
    rs = np.random.RandomState(0)
    beta = np.stack([x, x[::-1].copy(), x + 0.01 * rs.randn(*x.shape)])
    times = np.array([1.0, 2.0, 6.0])
    r = experiment.summarize(beta, times)
    avg = beta.mean(axis=0)
    ESS = tools.CalculateESS(avg, avg.shape[0] - 1)
    assert np.array_equal(r["avg_beta_posterior"], avg) and r["avg_time_taken"] == 3.0 and np.array_equal(r["ESS"], ESS)
    assert (r["Min"], r["Median"], r["Mean"], r["Max"]) == (ESS.min(), np.median(ESS), ESS.mean(), ESS.max())
    assert r["Time"] == 3.0 and r["Time per Min ESS"] == round(3.0 / ESS.min(), 6)
    assert np.allclose(r["ESS_per_run"][0], g["ess"], rtol=1e-9)          # run 0 IS the reference's chain: its ESS is the golden one
    assert r["per_run"]["Min"] == np.mean([e.min() for e in r["ESS_per_run"]])
    buf = io.StringIO(); experiment.report(r, file=buf)
    assert buf.getvalue().splitlines()[0] == "ESS" and buf.getvalue().splitlines()[-1].startswith("Time per Min ESS:")


def test_runs_are_the_same_batched_or_sequential(oracle):
    d = np.load(os.path.join(GOLDEN, "data_pima.npz"))
    kw = dict(NumOfIterations=30, BurnIn=10, _lib=oracle)
    a = experiment.run_experiment(d["XX"], d["t"], "RMHMC", n_experiments=3, batched=False, seed=5, **kw)
    b = experiment.run_experiment(d["XX"], d["t"], "RMHMC", n_experiments=3, batched=True, seed=5, **kw)
    assert a["results_beta"].shape == (3, 20, d["XX"].shape[1]) and a["results_time"].shape == (3,)
    assert np.array_equal(a["results_beta"], b["results_beta"])             # run i = chain i of the seed, however it is executed
    assert not np.array_equal(a["results_beta"][0], a["results_beta"][1])   # and the runs are different chains
    assert np.allclose(a["ESS"], b["ESS"]) and a["Min"] > 0
    h = experiment.run_experiment(d["XX"], d["t"], "HMC", n_experiments=2, batched=True, seed=1, NumOfIterations=20, BurnIn=5, _lib=oracle)
    assert h["results_beta"].shape == (2, 15, d["XX"].shape[1]) and h["sampler"] == "HMC"

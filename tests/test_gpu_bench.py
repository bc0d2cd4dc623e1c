"""bench.py and the device-resident write-out on the GPU box: the multi-rank code path of bench.py rehearsed with two ranks sharing
the one GPU (gloo for the collectives, exactly the calls the nccl run makes), and the `_dev` entry points of include/rmhmc.h against
their host twins.  Needs an MI355X: run with  pytest -m gpu."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT
from riemannhamiltonianmontecarlo_amd import _capi
from riemannhamiltonianmontecarlo_amd.data import synthetic_logreg

pytestmark = pytest.mark.gpu


def _run_bench(extra, nproc=1, env_extra=None, timeout=600):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **(env_extra or {}))
    cmd = [sys.executable]
    if nproc > 1:
        cmd += ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
                "--master-port", "29631"]
    cmd += [os.path.join(ROOT, "bench.py"), "--gpus", str(nproc)] + extra
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]          # rank 0 prints ONE JSON line
    return json.loads(lines[0])


def test_bench_two_ranks_gloo():
    """`bench.py --gpus 2` under torch.distributed.run with BENCH_BACKEND=gloo: both ranks run their shard (global chain ids
    rank*n ...), the barrier / max-over-ranks timing and the write-out gather see 2 ranks, rank 0 reports the whole job."""
    out = _run_bench(["--steps", "4", "--warmup", "1", "--workload", "c2", "--ess-iters", "20", "--no-cpu-baseline", "--no-alternates"],
                     nproc=2, env_extra={"BENCH_BACKEND": "gloo"})
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["steps"] == 4
    assert out["config"]["chains_total"] == 2 * out["config"]["chains_per_gpu"] == 2048
    assert out["gathered_chains"] == 2048 and out["all_finite"]
    assert out["value"] > 0 and abs(out["value"] - 2048 * 4 / (out["ms_per_step"] * 4e-3)) < 1e-6 * out["value"]
    assert out["min_ess"]["chains"] == 2048 and out["min_ess"]["min_ess_per_sec"] > 0
    assert "cpu_baseline" not in out


def test_bench_default_line_shape():
    """The N = 1 line on a small workload: every object the measurement contract names is present and self-consistent."""
    out = _run_bench(["--steps", "3", "--warmup", "1", "--workload", "c1", "--ess-iters", "30"])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline", "min_ess", "min_ess_per_sec"):
        assert k in out, k
    assert out["vs_baseline"] is None and out["dtype"] == "f64" and "workload" in out["config"] and "model" not in out["config"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in out["roofline"], k
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in out["cpu_baseline"], k
    assert out["cpu_baseline"]["kind"] == "port" and out["cpu_baseline"]["reference_algorithm"]["kind"] == "port"


def test_device_resident_outputs_match_host_outputs(hip):
    """rmhmc_chains_state_dev / rmhmc_sample_dev / rmhmc_sample_stats_dev write the same numbers into device tensors that their host
    twins copy out (same seed, same streams): what RCCL gathers is what the host API returns."""
    M, D, n = 700, 12, 70
    XX, t = synthetic_logreg(M, D, 3)
    dev = torch.device("cuda", 0)
    with hip.context(M, D, n, flags=0) as ctx:
        ctx.set_data(XX, t)
        ctx.chains_init(seed=5, L=4, eps=0.5, K=4)
        ctx.chains_run(9)
        w, it, acc = ctx.chains_state()
        wd, itd, accd = ctx.chains_state_dev(dev)
        assert wd.is_cuda and np.array_equal(wd.cpu().numpy(), w) and np.array_equal(itd.cpu().numpy(), it) and np.array_equal(accd.cpu().numpy(), acc)
        s, a, st, _ = ctx.sample(30, 10, 4, 0.5, 4, seed=11, chain_offset=3)
        sd, ad, std_, _ = ctx.sample_dev(dev, 30, 10, 4, 0.5, 4, seed=11, chain_offset=3)
        assert sd.shape == (n, 20, D) and np.array_equal(sd.cpu().numpy(), s)
        assert np.array_equal(ad.cpu().numpy(), a) and np.array_equal(std_.cpu().numpy(), st)
        h = ctx.sample_stats(30, 10, 4, 0.5, 4, seed=11, chain_offset=3)
        d = ctx.sample_stats_dev(dev, 30, 10, 4, 0.5, 4, seed=11, chain_offset=3)
        for k in ("mean", "var", "ess", "accepted", "leapfrog_steps"):
            assert np.array_equal(d[k].cpu().numpy(), h[k], equal_nan=True), k
        assert np.allclose(h["mean"], s.mean(axis=1), rtol=1e-12, atol=1e-14)

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


def test_bench_four_ranks_gloo_small_shards():
    """The rank count rehearsed beyond two (the GPU box allows at most 6 processes on its card, so 4 ranks stand in for config 4's 8):
    `bench.py --gpus 4 --chains 128`, gloo collectives, every rank's shard on the one GPU; global chain ids rank*128.., one gather."""
    out = _run_bench(["--steps", "3", "--warmup", "1", "--workload", "c2", "--chains", "128", "--ess-iters", "10", "--no-cpu-baseline",
                      "--no-alternates", "--option", "inflight=16"], nproc=4, env_extra={"BENCH_BACKEND": "gloo"})
    assert out["n_gpus"] == 4 and out["config"]["chains_total"] == 512 and out["gathered_chains"] == 512 and out["all_finite"]
    assert out["config"]["options"]["inflight"] == 16 and out["config"]["options"]["graph"] == 1
    assert out["min_ess"]["chains"] == 512


def _sharded_worker(rank, world, port, outfile, gather):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from riemannhamiltonianmontecarlo_amd.multi_gpu import sample_sharded
    dist.init_process_group("gloo", rank=rank, world_size=world)
    XX, t = synthetic_logreg(300, 12, 4)
    # the product library (default lib) under a gloo process group: outputs are written in HBM and hop to the host for the gather
    out = sample_sharded(XX if rank == 0 else None, t if rank == 0 else None, 7, NumOfIterations=24, BurnIn=8, seed=5, compat=False,
                         gather=gather)
    if rank == 0:
        payload, secs, info = out
        if gather == "samples":
            np.savez(outfile, samples=payload, acc=info["accepted"], steps=info["leapfrog_steps"])
        else:
            np.savez(outfile, **payload)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("gather", ["samples", "summary"])
def test_sample_sharded_hip_library_under_gloo(hip, tmp_path, gather):
    """sample_sharded with the HIP library and a non-nccl backend handed host tensors to the _dev entry points.  Three
    ranks (a ragged split of 7 chains: 3 + 2 + 2) on the one GPU, data broadcast from rank 0: bit-equal to one context."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / "r0.npz")
    mp.spawn(_sharded_worker, args=(3, port, out, gather), nprocs=3, join=True)
    XX, t = synthetic_logreg(300, 12, 4)
    with hip.context(300, 12, 7, flags=0) as ctx:
        ctx.set_data(XX, t)
        smp, acc, steps, _ = ctx.sample(24, 8, seed=5)
    g = np.load(out)
    if gather == "samples":
        assert np.array_equal(g["samples"], smp) and np.array_equal(g["acc"], acc) and np.array_equal(g["steps"], steps)
    else:
        assert np.allclose(g["mean"], smp.mean(1), rtol=1e-12, atol=1e-14) and g["min_ess"].shape == (7,)


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
    assert out["cpu_baseline"]["reference_numpy"]["value"] > 0          # the NumPy restatement of the reference's literal algorithm
    assert out["config"]["options"]["graph"] == 1 and out["config"]["options"]["inflight"] == 32


def test_bench_checkpoint_resume_is_the_state_the_burn_in_leaves(tmp_path):
    """tools/profile.sh starts every profiled pass from `bench.py --save-state`: the resumed run continues the chains the burn-in left
    (rmhmc_chains_state is a complete checkpoint; a chain stopped in mid-trajectory replays that transition from its start, so the
    counters after the same number of further global steps differ by at most that one transition per chain)."""
    common = ["--steps", "6", "--warmup", "2", "--workload", "c2", "--chains", "256", "--burn-in-steps", "24", "--ess-iters", "0",
              "--no-cpu-baseline", "--no-alternates"]
    a = _run_bench(common)
    ck = str(tmp_path / "ck.npz")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + common + ["--save-state", ck], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and os.path.exists(ck), r.stderr[-1000:]
    b = _run_bench(common + ["--load-state", ck, "--no-graph"])
    assert a["all_finite"] and b["all_finite"] and abs(a["acceptance_rate"] - b["acceptance_rate"]) < 0.05
    assert b["config"]["options"]["graph"] == 0


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


@pytest.mark.parametrize("shape", ["medium", "fused", "generic"])
def test_progress_reports_follow_the_reference_schedule(hip, capsys, shape):
    """rmhmc_set_progress: the reference prints '<k*50> iterations completed.' / 'Acceptance: ...' at the top of every iteration whose
    number+1 is a multiple of 50 (rmhmc.py:38-45: 49 proposals in the first window, 50 afterwards) and the burn-in banner after iteration
    BurnIn (:194-196).  The reports cut the run at those marks; the samples must not depend on them (one-launch, fused and generic
    stepping paths)."""
    from riemannhamiltonianmontecarlo_amd import RMHMC
    M, D, n = {"medium": (400, 12, 1), "fused": (300, 6, 5), "generic": (500, 40, 3)}[shape]
    XX, t = synthetic_logreg(M, D, 2)
    events = []
    with hip.context(M, D, n, flags=_capi.COMPAT) as ctx:
        ctx.set_data(XX, t)
        ctx.set_progress(lambda ev, it, acc, itot: events.append((ev, it, acc, itot)))
        a = ctx.sample(230, 120, 6, 0.5, 4, seed=3)
        ctx.set_progress(None)
        b = ctx.sample(230, 120, 6, 0.5, 4, seed=3)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    assert [e[1] for e in events if e[0] == _capi.EV_BURNIN_DONE] == [121]
    assert [e[3] for e in events if e[0] == _capi.EV_BURNIN_DONE] == [121 * n]         # every chain at exactly BurnIn + 1 transitions
    prog = [e for e in events if e[0] == _capi.EV_PROGRESS]
    if n == 1:   # one chain: cut at every milestone, exact counters, banner between the reports where the reference prints it
        assert [e[1] for e in prog] == [49, 99, 149, 199] and [e[3] for e in prog] == [49, 99, 149, 199]
        assert [e[1] for e in events] == [49, 99, 121, 149, 199]
    else:        # several chains: nobody waits; a report names the largest milestone the slowest chain has passed
        ms = [e[1] for e in prog]
        assert ms == sorted(set(ms)) and set(ms) <= {49, 99, 149, 199} and ms[-1] == 199
        assert all(e[3] >= n * e[1] for e in prog)                                     # the other chains are ahead
    accs = [e[2] for e in events]
    assert accs == sorted(accs) and accs[-1] <= a[1].sum() <= 230 * n                 # accepted-so-far is monotone and consistent with the total
    assert all(e[2] <= e[3] for e in events)
    if n == 1:
        capsys.readouterr()
        RMHMC(XX, t, NumOfIterations=230, BurnIn=120, seed=3)
        out = capsys.readouterr().out.splitlines()
        assert out[0] == "50 iterations completed." and out[1] == "Acceptance: %s" % (events[0][2] / 49.0)
        assert out[2] == "100 iterations completed." and out[3] == "Acceptance: %s" % ((events[1][2] - events[0][2]) / 50.0)
        assert out[4] == "Burn-in complete, now drawing posterior samples."
        assert out[5] == "150 iterations completed." and out[-1].startswith("Time drawing posterior: ")


_SHARDED_SCRIPT = r"""
import os, sys, json
import numpy as np
sys.path.insert(0, sys.argv[1])
import torch, torch.distributed as dist
from riemannhamiltonianmontecarlo_amd import _capi, multi_gpu
from riemannhamiltonianmontecarlo_amd.data import synthetic_logreg
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29647")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1"); os.environ.setdefault("LOCAL_RANK", "0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))            # RCCL
XX, t = synthetic_logreg(600, 10, 4)
smp, secs, info = multi_gpu.sample_sharded(XX, t, 37, 40, 15, seed=6, compat=False, gather="samples")
summ, _, info2 = multi_gpu.sample_sharded(XX, t, 37, 40, 15, seed=6, compat=False, gather="summary")
lib = _capi.load_hip_library()
with lib.context(600, 10, 37, flags=0) as ctx:
    ctx.set_data(XX, t)
    ref, acc, steps, _ = ctx.sample(40, 15, seed=6)
ok = bool(np.array_equal(smp, ref) and np.array_equal(info["accepted"], acc) and np.array_equal(info["leapfrog_steps"], steps)
          and np.allclose(summ["mean"], ref.mean(axis=1), rtol=1e-12, atol=1e-14) and np.array_equal(info2["accepted"], acc))
print(json.dumps({"ok": ok, "backend": dist.get_backend(), "world": dist.get_world_size(), "shape": list(smp.shape)}))
dist.destroy_process_group()
"""


def test_sharded_sampler_gathers_from_hbm_through_rccl(tmp_path):
    """multi_gpu.sample_sharded with the nccl backend (= RCCL) on the one GPU of the box: the sampler's device-resident outputs go
    through torch.distributed.gather as CUDA tensors and come out bit-equal to a direct rmhmc_sample call."""
    script = tmp_path / "sharded.py"
    script.write_text(_SHARDED_SCRIPT)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, str(script), ROOT], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out == {"ok": True, "backend": "nccl", "world": 1, "shape": [37, 25, 10]}


def test_bench_one_rank_rccl():
    """bench.py's multi-rank branch with the REAL backend: one rank under torch.distributed.run with BENCH_FORCE_DIST=1 initialises RCCL
    ("nccl"), runs the barriers, the max-over-ranks all-reduce and the write-out gather on CUDA tensors produced by rmhmc_chains_state_dev
    and rmhmc_sample_stats_dev."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", BENCH_FORCE_DIST="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29653", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "1", "--workload", "c2",
           "--ess-iters", "20", "--no-cpu-baseline", "--no-alternates"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["n_gpus"] == 1 and out["gathered_chains"] == 1024 and out["all_finite"] and out["min_ess"]["min_ess_per_sec"] > 0


def test_experiment_driver_on_the_gpu(hip):
    """experiment.run_experiment (main.py:43-79) through the HIP library: run i is chain i of the seed whether the runs are executed one
    after another (as main.py does) or as one batch; the summary carries the reference's quantities."""
    from conftest import GOLDEN
    from riemannhamiltonianmontecarlo_amd import experiment, tools
    d = np.load(os.path.join(GOLDEN, "data_heart.npz"))
    kw = dict(NumOfIterations=260, BurnIn=60, compat=False)
    a = experiment.run_experiment(d["XX"], d["t"], "RMHMC", n_experiments=3, batched=False, seed=11, **kw)
    b = experiment.run_experiment(d["XX"], d["t"], "RMHMC", n_experiments=3, batched=True, seed=11, **kw)
    assert a["results_beta"].shape == (3, 200, 14) and np.array_equal(a["results_beta"], b["results_beta"])
    assert a["results_time"].shape == (3,) and (a["results_time"] > 0).all() and np.all(b["results_time"] == b["results_time"][0])
    assert np.array_equal(a["ESS"], tools.CalculateESS(a["results_beta"].mean(axis=0), 199))
    assert a["Time per Min ESS"] == round(a["avg_time_taken"] / a["ESS"].min(), 6)
    assert 20 < a["per_run"]["Min"] <= 200            # RMHMC on heart: nearly independent samples (paper Table 6: 4862 of 5000)


def test_hmc_shim_prints_like_the_reference(hip, capsys):
    """hmc.py:85-97: after iterations 0, 50, ... up to BurnIn '<number> iterations completed.' and the acceptance rate of the window, the
    burn-in banner after iteration BurnIn, the time at the end."""
    from riemannhamiltonianmontecarlo_amd import HMC
    XX, t = synthetic_logreg(300, 6, 1)
    capsys.readouterr()
    w, secs = HMC(XX, t, NumOfIterations=130, BurnIn=60, NumOfLeapFrogSteps=20, StepSize=0.05, seed=2)
    out = capsys.readouterr().out.splitlines()
    assert w.shape == (70, 6) and secs > 0
    assert out[0] == "0 iterations completed." and out[1] in ("Acceptance: 1.0", "Acceptance: 0.0")
    assert out[2] == "50 iterations completed." and out[3].startswith("Acceptance: ") and 0.0 <= float(out[3].split()[1]) <= 1.0
    assert out[4] == "Burn-in complete, now drawing posterior samples."
    assert len(out) == 6 and out[5].startswith("Time drawing posterior: ")
    # BurnIn a multiple of 50 (the default 1000 is one): the report of iteration BurnIn comes just before the banner (hmc.py:85-94)
    HMC(XX, t, NumOfIterations=120, BurnIn=50, NumOfLeapFrogSteps=20, StepSize=0.05, seed=2)
    out = capsys.readouterr().out.splitlines()
    assert [o for o in out if not o.startswith("Acceptance")][:3] == ["0 iterations completed.", "50 iterations completed.",
                                                                      "Burn-in complete, now drawing posterior samples."]
    assert len(out) == 6
    # RMHMC with BurnIn % 50 == 48: the print of iteration BurnIn + 1 follows the banner (rmhmc.py:38 prints at the top of the iteration)
    from riemannhamiltonianmontecarlo_amd import RMHMC
    RMHMC(XX, t, NumOfIterations=110, BurnIn=48, seed=2)
    out = [o for o in capsys.readouterr().out.splitlines() if not o.startswith("Acceptance")]
    assert out[:3] == ["Burn-in complete, now drawing posterior samples.", "50 iterations completed.", "100 iterations completed."]

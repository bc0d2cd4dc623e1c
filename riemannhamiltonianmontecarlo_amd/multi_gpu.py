"""Chain sharding across the GPUs of one node.

Chains never interact (the reference has exactly one chain, ``rmhmc.py:37-191``), so the path
partitions by chain index: rank r runs chains ``[start_r, end_r)`` on its own GPU with X, t replicated by
its own ``rmhmc_set_data`` and **no collective inside the sampling loop**.  The Philox counters use the
global chain id (``chain_offset``), so the samples do not depend on the number of ranks.  The single
exchange is the gather at write-out (``torch.distributed``: RCCL over xGMI for the ``nccl`` backend, gloo
in the CPU tests).  One process per GPU, launched by ``torch.distributed.run``.
"""
import os

import numpy as np

from . import _capi


def shard_range(n_total, world, rank):
    """Contiguous, balanced partition of chain ids: the first (n_total % world) ranks get one more."""
    base, extra = divmod(int(n_total), int(world))
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def _dist():
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        raise RuntimeError("torch.distributed is not initialised (launch with torch.distributed.run)")
    return dist


def _gather_rows(local, counts, device):
    """Gather variable-length leading-dimension arrays to rank 0 (padding to the longest shard)."""
    import torch
    dist = _dist()
    rank, world = dist.get_rank(), dist.get_world_size()
    mx = max(counts)
    pad = np.zeros((mx,) + local.shape[1:], dtype=local.dtype)
    pad[: local.shape[0]] = local
    tl = torch.from_numpy(pad).to(device)
    bufs = [torch.empty_like(tl) for _ in range(world)] if rank == 0 else None
    dist.gather(tl, bufs, dst=0)
    if rank != 0:
        return None
    return np.concatenate([b.cpu().numpy()[: counts[r]] for r, b in enumerate(bufs)], axis=0)


def sample_sharded(XX, t, n_chains, NumOfIterations=6000, BurnIn=1000, NumOfLeapFrogSteps=6, StepSize=0.5,
                   NumOfNewtonSteps=4, *, seed=0, compat=True, theta0=None, alpha=100.0, gather="samples", lib=None):
    """Run ``n_chains`` chains sharded over the ranks of the initialised process group.

    gather="samples": rank 0 returns (samples [n_chains,S,D], seconds, info); other ranks return None.
    gather="summary": only per-chain posterior mean / variance / min-ESS travel (config 4 at S=5000 would be
    168 GB of raw samples); rank 0 returns (summary dict, seconds, info).
    ``lib`` is the loaded C-ABI library (default: the HIP library; the CPU tests inject the oracle).
    """
    import torch
    dist = _dist()
    rank, world = dist.get_rank(), dist.get_world_size()
    backend = dist.get_backend()
    local_rank = int(os.environ.get("LOCAL_RANK", rank))
    device = torch.device("cuda", local_rank) if backend == "nccl" else torch.device("cpu")
    lib = lib if lib is not None else _capi.load_hip_library()
    XX = np.ascontiguousarray(XX, dtype=np.float64)
    N, D = XX.shape
    start, end = shard_range(n_chains, world, rank)
    counts = [shard_range(n_chains, world, r)[1] - shard_range(n_chains, world, r)[0] for r in range(world)]
    n_local = end - start
    S = NumOfIterations - BurnIn
    summ = None
    if n_local > 0:
        th = None if theta0 is None else np.broadcast_to(theta0, (n_chains, D))[start:end]
        with lib.context(N, D, n_local, flags=(_capi.COMPAT if compat else 0) | _capi.auto_metric_flags(D, n_local, M=N), device=local_rank if backend == "nccl" else 0) as ctx:
            ctx.set_data(XX, t, alpha)
            if gather == "samples":
                smp, acc, steps, secs = ctx.sample(NumOfIterations, BurnIn, NumOfLeapFrogSteps, StepSize, NumOfNewtonSteps,
                                                   seed=seed, chain_offset=start, theta0=th)
            else:  # reduced on the device: no sample transfer at all (rmhmc_sample_stats)
                st = ctx.sample_stats(NumOfIterations, BurnIn, NumOfLeapFrogSteps, StepSize, NumOfNewtonSteps, seed=seed,
                                      chain_offset=start, theta0=th)
                acc, steps, secs = st["accepted"], st["leapfrog_steps"], st["seconds"]
                summ = np.concatenate([st["mean"], st["var"], np.nanmin(st["ess"], axis=1, keepdims=True)], axis=1)
                smp = np.zeros((n_local, 0, D))
    else:
        smp = np.zeros((0, S, D)); acc = np.zeros(0, dtype=np.int64); steps = np.zeros(0, dtype=np.int64); secs = 0.0
        summ = np.zeros((0, 2 * D + 1))
    # timing: the job is as slow as its slowest rank
    tt = torch.tensor([secs], dtype=torch.float64, device=device)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    seconds = float(tt.item())
    acc_all = _gather_rows(acc.reshape(-1, 1), counts, device)
    steps_all = _gather_rows(steps.reshape(-1, 1), counts, device)
    if gather == "samples":
        payload = _gather_rows(smp, counts, device)
    else:
        g = _gather_rows(summ, counts, device)
        payload = None if g is None else dict(mean=g[:, :D], var=g[:, D:2 * D], min_ess=g[:, 2 * D])
    if rank != 0:
        return None
    info = dict(accepted=acc_all.ravel(), leapfrog_steps=steps_all.ravel(), world=world, counts=counts)
    return payload, seconds, info

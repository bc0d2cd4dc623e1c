"""Chain sharding across the GPUs of one node.

Chains never interact (the reference has exactly one chain, ``rmhmc.py:37-191``), so the path
partitions by chain index: rank r runs chains ``[start_r, end_r)`` on its own GPU with X, t replicated by
its own ``rmhmc_set_data`` (ranks that are not handed the data get it from rank 0 by one broadcast before the
run) and **no collective inside the sampling loop**.  The Philox counters use the
global chain id (``chain_offset``), so the samples do not depend on the number of ranks.  The single
exchange is the gather at write-out (``torch.distributed``: RCCL over xGMI for the ``nccl`` backend, gloo
in the CPU tests), fed from device memory: the sampler writes its outputs through the ``_dev`` entry points of
include/rmhmc.h into torch tensors in HBM and those tensors are what RCCL sends.  One process per GPU, launched by ``torch.distributed.run``.
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


def _gather_rows(local, counts):
    """Gather variable-length leading-dimension TENSORS to rank 0, padding to the longest shard.  `local` lives where the
    collective runs: in HBM for the nccl backend (RCCL moves it over xGMI straight from the buffer the sampler wrote - no host
    bounce), in host memory for gloo.  Rank 0 gets one tensor on the same device, other ranks None."""
    import torch
    dist = _dist()
    rank, world = dist.get_rank(), dist.get_world_size()
    mx = max(counts)
    if local.shape[0] != mx:
        pad = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        pad[: local.shape[0]] = local
        local = pad
    local = local.contiguous()
    bufs = [torch.empty_like(local) for _ in range(world)] if rank == 0 else None
    dist.gather(local, bufs, dst=0)
    if rank != 0:
        return None
    return torch.cat([b[: counts[r]] for r, b in enumerate(bufs)], dim=0)


def _replicate_data(XX, t, device):
    """Every rank needs (XX, t) for its own rmhmc_set_data.  Ranks that pass None receive rank 0's arrays by one broadcast (5 MB at
    config 4) before the run; when every rank has the data already (synthetic recipes) nothing is sent."""
    import torch
    dist = _dist()
    have = torch.tensor([0 if XX is None else 1], dtype=torch.int64, device=device)
    dist.all_reduce(have, op=dist.ReduceOp.MIN)
    if int(have.item()) == 1:
        return np.ascontiguousarray(XX, dtype=np.float64), np.ascontiguousarray(t, dtype=np.float64).reshape(-1)
    shape = torch.zeros(2, dtype=torch.int64, device=device)
    if dist.get_rank() == 0:
        if XX is None:
            raise ValueError("rank 0 must hold the data")
        XX = np.ascontiguousarray(XX, dtype=np.float64)
        shape = torch.tensor(XX.shape, dtype=torch.int64, device=device)
    dist.broadcast(shape, src=0)
    N, D = int(shape[0]), int(shape[1])
    buf = torch.empty(N * D + N, dtype=torch.float64, device=device)
    if dist.get_rank() == 0:
        buf.copy_(torch.from_numpy(np.concatenate([XX.ravel(), np.asarray(t, dtype=np.float64).ravel()])))
    dist.broadcast(buf, src=0)
    h = buf.cpu().numpy()
    return np.ascontiguousarray(h[: N * D].reshape(N, D)), np.ascontiguousarray(h[N * D:])


def nanmin_rows(ess):
    """min over the dimensions of a (chains, D) ESS tensor ignoring NaN entries (a constant coordinate has no ESS); a chain whose ESS is NaN
    in EVERY dimension (a constant or diverged chain) gets NaN - not +inf, which would read as a perfect chain in any sum or mean"""
    import torch
    m = torch.nan_to_num(ess, nan=float("inf")).amin(dim=1, keepdim=True)
    m[torch.isinf(m)] = float("nan")
    return m


def sample_sharded(XX, t, n_chains, NumOfIterations=6000, BurnIn=1000, NumOfLeapFrogSteps=6, StepSize=0.5,
                   NumOfNewtonSteps=4, *, seed=0, compat=True, theta0=None, alpha=100.0, gather="samples", lib=None, options=None):
    """Run ``n_chains`` chains sharded over the ranks of the initialised process group.

    gather="samples": rank 0 returns (samples [n_chains,S,D], seconds, info); other ranks return None.
    gather="summary": only per-chain posterior mean / variance / min-ESS travel (config 4 at S=5000 would be
    168 GB of raw samples); rank 0 returns (summary dict, seconds, info).  min_ess of a chain whose ESS is NaN in
    every dimension (a constant or diverged chain) is NaN, not a number.
    XX, t may be None on ranks other than 0: they are broadcast once from rank 0.
    ``lib`` is the loaded C-ABI library (default: the HIP library; the CPU tests inject the oracle); ``options``: tuning options of
    include/rmhmc.h for the contexts.
    """
    import torch as _t
    dist = _dist()
    rank, world = dist.get_rank(), dist.get_world_size()
    backend = dist.get_backend()
    local_rank = int(os.environ.get("LOCAL_RANK", rank))
    lib = lib if lib is not None else _capi.load_hip_library()
    # where the collective runs (HBM for RCCL, host memory for gloo) and where the sampler writes its outputs: the HIP library always
    # writes through device pointers of its own GPU, whatever the backend; with gloo those tensors take one hop to the host first
    cdev = _t.device("cuda", local_rank) if backend == "nccl" else _t.device("cpu")
    gpu = (local_rank % max(1, _t.cuda.device_count())) if lib.on_gpu else 0
    odev = _t.device("cuda", gpu) if lib.on_gpu else _t.device("cpu")
    XX, t = _replicate_data(XX, t, cdev)
    N, D = XX.shape
    start, end = shard_range(n_chains, world, rank)
    counts = [shard_range(n_chains, world, r)[1] - shard_range(n_chains, world, r)[0] for r in range(world)]
    n_local = end - start
    S = NumOfIterations - BurnIn
    if n_local > 0:
        th = None if theta0 is None else np.broadcast_to(theta0, (n_chains, D))[start:end]
        with lib.context(N, D, n_local, flags=(_capi.COMPAT if compat else 0) | _capi.auto_metric_flags(D, n_local, M=N), device=gpu,
                         options=options) as ctx:
            ctx.set_data(XX, t, alpha)
            # device-resident write-out (rmhmc_sample_dev / rmhmc_sample_stats_dev): the outputs stay in this rank's HBM
            if gather == "samples":
                smp, acc, steps, secs = ctx.sample_dev(odev, NumOfIterations, BurnIn, NumOfLeapFrogSteps, StepSize, NumOfNewtonSteps,
                                                       seed=seed, chain_offset=start, theta0=th)
                summ = None
            else:  # reduced on the device: no raw sample leaves the GPU at all
                st = ctx.sample_stats_dev(odev, NumOfIterations, BurnIn, NumOfLeapFrogSteps, StepSize, NumOfNewtonSteps, seed=seed,
                                          chain_offset=start, theta0=th)
                acc, steps, secs = st["accepted"], st["leapfrog_steps"], st["seconds"]
                summ = _t.cat([st["mean"], st["var"], nanmin_rows(st["ess"])], dim=1)
                smp = None
    else:
        smp = _t.zeros((0, S, D), dtype=_t.float64, device=odev)
        acc = _t.zeros(0, dtype=_t.int64, device=odev); steps = _t.zeros(0, dtype=_t.int64, device=odev); secs = 0.0
        summ = _t.zeros((0, 2 * D + 1), dtype=_t.float64, device=odev)
    # timing: the job is as slow as its slowest rank
    tt = _t.tensor([secs], dtype=_t.float64, device=cdev)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    seconds = float(tt.item())
    # the ONE exchange of the sharded path: counters and payload gathered to rank 0 from where the sampler left them
    cnt_all = _gather_rows(_t.stack([acc, steps], dim=1).to(cdev), counts)
    payload = _gather_rows((smp if gather == "samples" else summ).to(cdev), counts)
    if rank != 0:
        return None
    cnt_all = cnt_all.cpu().numpy()
    if gather == "samples":
        payload = payload.cpu().numpy()   # the drop-in contract returns host arrays; the transfer happens once, on rank 0
    else:
        g = payload.cpu().numpy()
        payload = dict(mean=g[:, :D], var=g[:, D:2 * D], min_ess=g[:, 2 * D])
    info = dict(accepted=cnt_all[:, 0], leapfrog_steps=cnt_all[:, 1], world=world, counts=counts)
    return payload, seconds, info

"""World-size-2 and world-size-8 tests of the sharded path on CPU (gloo), with ragged and empty shards.  The compute engine is injected: here the CPU
oracle stands in for the HIP library so that the host-side sharding / chain_offset / gather logic is
exercised without a GPU (on the GPU box the same function runs with the HIP library and nccl)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

from conftest import GOLDEN, ORACLE_LIB, ROOT
from riemannhamiltonianmontecarlo_amd.multi_gpu import shard_range


def test_nanmin_rows_keeps_nan_for_a_chain_without_any_finite_ess():
    import torch
    from riemannhamiltonianmontecarlo_amd.multi_gpu import nanmin_rows
    e = torch.tensor([[3.0, float("nan"), 2.0], [float("nan")] * 3, [5.0, 7.0, 6.0]], dtype=torch.float64)
    m = nanmin_rows(e).ravel()
    assert m[0] == 2.0 and torch.isnan(m[1]) and m[2] == 5.0


def test_shard_range_partitions():
    for n, w in ((10, 4), (8192, 8), (3, 8), (65536, 8), (7, 1)):
        r = [shard_range(n, w, k) for k in range(w)]
        assert r[0][0] == 0 and r[-1][1] == n
        assert all(r[k][1] == r[k + 1][0] for k in range(w - 1))
        sizes = [b - a for a, b in r]
        assert max(sizes) - min(sizes) <= 1


def _worker(rank, world, port, outfile, gather, n_chains=5, bcast=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      OMP_NUM_THREADS="2")
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from riemannhamiltonianmontecarlo_amd import _capi
    from riemannhamiltonianmontecarlo_amd.multi_gpu import sample_sharded
    dist.init_process_group("gloo", rank=rank, world_size=world)
    d = np.load(os.path.join(GOLDEN, "data_pima.npz"))
    lib = _capi.RmhmcLib(ORACLE_LIB)
    XX, t = (d["XX"], d["t"]) if (rank == 0 or not bcast) else (None, None)   # bcast: only rank 0 holds the data
    out = sample_sharded(XX, t, n_chains, NumOfIterations=30, BurnIn=10, seed=17, gather=gather, lib=lib)
    if rank == 0:
        payload, secs, info = out
        assert info["world"] == world and sum(info["counts"]) == n_chains
        if gather == "samples":
            np.savez(outfile, samples=payload, secs=secs, acc=info["accepted"], steps=info["leapfrog_steps"])
        else:
            np.savez(outfile, mean=payload["mean"], var=payload["var"], min_ess=payload["min_ess"], secs=secs)
    else:
        assert out is None
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


@pytest.mark.parametrize("gather", ["samples", "summary"])
def test_two_ranks_equal_one_rank(oracle, tmp_path, gather):
    out = str(tmp_path / "r0.npz")
    mp.spawn(_worker, args=(2, _free_port(), out, gather), nprocs=2, join=True)
    d = np.load(os.path.join(GOLDEN, "data_pima.npz"))
    with oracle.context(d["XX"].shape[0], d["XX"].shape[1], 5) as ctx:
        ctx.set_data(d["XX"], d["t"])
        s, acc, steps, _ = ctx.sample(30, 10, seed=17)
    g = np.load(out)
    if gather == "samples":
        assert np.array_equal(g["samples"], s)      # sharding does not change a single bit
        assert np.array_equal(g["acc"], acc) and np.array_equal(g["steps"], steps)
    else:
        assert np.allclose(g["mean"], s.mean(1)) and np.allclose(g["var"], s.var(1))
        assert g["min_ess"].shape == (5,) and (g["min_ess"] > 0).all()
    assert g["secs"] > 0


@pytest.mark.parametrize("gather,n_chains", [("samples", 13), ("summary", 13), ("samples", 5), ("summary", 5)])
def test_eight_ranks_ragged_and_empty_shards(oracle, tmp_path, gather, n_chains):
    """Config 4's rank count: 13 chains over 8 ranks (shards of 2 and 1), 5 chains over 8 ranks (three ranks hold nothing and still take
    part in the two gathers), the data broadcast from rank 0: bit-equal to one rank."""
    out = str(tmp_path / "r0.npz")
    mp.spawn(_worker, args=(8, _free_port(), out, gather, n_chains, True), nprocs=8, join=True)
    d = np.load(os.path.join(GOLDEN, "data_pima.npz"))
    with oracle.context(d["XX"].shape[0], d["XX"].shape[1], n_chains) as ctx:
        ctx.set_data(d["XX"], d["t"])
        s, acc, steps, _ = ctx.sample(30, 10, seed=17)
    g = np.load(out)
    if gather == "samples":
        assert np.array_equal(g["samples"], s)
        assert np.array_equal(g["acc"], acc) and np.array_equal(g["steps"], steps)
    else:
        assert np.allclose(g["mean"], s.mean(1)) and np.allclose(g["var"], s.var(1))
        assert g["min_ess"].shape == (n_chains,) and (g["min_ess"] > 0).all()

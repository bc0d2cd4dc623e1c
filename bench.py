#!/usr/bin/env python3
"""Benchmark of the RMHMC hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over the whole batch: every chain executes ONE generalised
leapfrog step (rmhmc.py:96-163), including the transition bookkeeping (momentum draw, Hamiltonians,
accept/reject) of the chains whose trajectory starts or ends on that step.  `value` is leapfrog
steps per second summed over all chains and GPUs, with X, t and the chain state resident in HBM.

Workloads (BASELINE.json configs):
  c3  8192 chains/GPU, D=64, M=10000  (default; the configuration the north-star target is quoted
      on; with --gpus 8 it is config 4: 65536 chains sharded 8192 per GPU, weak scaling)
  c2  1024 chains, D=8, M=1000
  c1  bundled australian data (M=690, D=15), 1 chain
  c5  4096 chains, D=256, M=50000 (large-D path)

The timed region runs the product path exactly as a user gets it (the global step replayed from a
hipGraph, no per-kernel events).  Right after it the SAME K steps are repeated once more with HIP events
around every launch on the library's stream ("kernel_seconds", un-timed repetition: events disable the
graph path); the "roofline" object is computed from that repetition.

Extra objects in the JSON line: "roofline" (dominant kernel = metric assembly), "roofline_fp64" (the
same workload and steps with the metric on the fp64 matrix cores: the same-arithmetic number and its
fraction of the fp64 MFMA peak), "min_ess" (the second half of BASELINE.json's metric: min-ESS/sec over
the TimeTaken window of rmhmc.py:194-198, ESS by tools.py:32-74) and "cpu_baseline" (the CPU oracle timed
on the host cores on a bounded sample; rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12        # B/s, MI355X_MICROARCH.md (spec; 6.3e12 achievable, 5.94e12 measured by tools/mfma_probe)
INT8_MFMA_PEAK = 5.0e15   # op/s dense int8 matrix (2x the bf16 rate, MI355X_MICROARCH.md)
INT8_MFMA_STREAM = 3.39e15  # op/s a bare v_mfma_i32_32x32x32_i8 stream sustains on random bytes with two waves per SIMD (2.89e15 with one; 5.04e15
                            # on zeros: the clock is given back under load; profiles/r02_i8_stream_probe.txt)
INT8_FEED_FREE_REAL = 2.84e15  # op/s of the library's own tile on the REAL operand planes of config 3 when nothing is staged after the first stages
                               # and no fragment is re-read (I8_ABLATE timing builds, profiles/r03_i8_real_ceiling.txt: 2.8-2.9e15 on the box that ran
                               # the product at 2.48e15)
FP64_MFMA_PEAK = 78.6e12  # flop/s dense fp64 matrix (spec); tools/mfma_probe measures 75.1e12

WORKLOADS = {
    "c3": dict(chains=8192, D=64, M=10000, desc="BASELINE config 3/4: 8192 chains per GPU, D=64, M=10000 synthetic logistic regression"),
    "c2": dict(chains=1024, D=8, M=1000, desc="BASELINE config 2: 1024 chains, D=8, M=1000 synthetic logistic regression"),
    "c1": dict(chains=1, D=15, M=690, desc="BASELINE config 1: bundled australian data, 1 chain"),
    "c5": dict(chains=4096, D=256, M=50000, desc="BASELINE config 5: 4096 chains, D=256, M=50000 (blocked Cholesky, 64-column blocks)"),
}


def load_problem(name):
    from riemannhamiltonianmontecarlo_amd.data import synthetic_logreg
    wl = WORKLOADS[name]
    if name == "c1":
        d = np.load(os.path.join(ROOT, "tests", "golden", "data_australian.npz"))
        return d["XX"], d["t"]
    return synthetic_logreg(wl["M"], wl["D"], 0)


def cpu_baseline(XX, t, flags, L, eps, K, budget_s=12.0, literal_budget_s=10.0):
    """Time the CPU oracle (oracle/librmhmc_oracle.so, the C restatement of rmhmc.py — kind "port") on
    the host cores with a bounded sample of the same workload: `cores` chains, a few global steps."""
    from riemannhamiltonianmontecarlo_amd import _capi
    import __graft_entry__ as ge
    if not os.path.exists(ge.ORACLE_LIB):
        return None
    oracle = _capi.RmhmcLib(ge.ORACLE_LIB)
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = min(avail, int(os.environ.get("RMHMC_CPU_THREADS", "16")))  # one GPU's CPU share on the box is 16 cores
    os.environ["OMP_NUM_THREADS"] = str(cores)
    M, D = XX.shape
    # grow the sample (steps up to 50, then chains) until it takes a good fraction of budget_s
    n, steps, dt = cores, 2, 0.0
    for _ in range(6):
        with oracle.context(M, D, n, flags=flags) as ctx:
            ctx.set_data(XX, t)
            ctx.chains_init(seed=1, L=L, eps=eps, K=K)
            t0 = time.perf_counter(); ctx.chains_run(steps); dt = time.perf_counter() - t0
        if dt >= budget_s / 3 or n >= 8192:
            break
        grow = min(16.0, budget_s / max(dt, 1e-6))
        new_steps = int(min(50, max(steps, steps * grow)))
        grow /= new_steps / steps
        steps = new_steps
        n = int(min(8192, max(n, int(n * grow) // cores * cores)))
    out = {"value": n * steps / dt, "unit": "leapfrog-steps/s", "cores": cores, "kind": "port",
           "sample": "%d chains x %d leapfrog steps of the same (M=%d, D=%d) workload, matrix-free C oracle, OpenMP over chains, %.1f s"
                     % (n, steps, M, D, dt)}
    # the same oracle on ONE core (one chain: the OpenMP loop over chains has a single iteration), SURVEY 8(d) asks for both
    st1 = max(2, int(steps * n / cores / 8))
    with oracle.context(M, D, 1, flags=flags) as ctx:
        ctx.set_data(XX, t)
        ctx.chains_init(seed=1, L=L, eps=eps, K=K)
        t0 = time.perf_counter(); ctx.chains_run(st1); dt1 = time.perf_counter() - t0
    out["one_core"] = {"value": st1 / dt1, "unit": "leapfrog-steps/s", "cores": 1, "kind": "port",
                       "sample": "1 chain x %d leapfrog steps, matrix-free C oracle, %.1f s" % (st1, dt1)}
    # the reference's own O(M D^3) formulation (forms the DxDxD tensor, LU inverse/solve: the literal variant of the
    # oracle, i.e. what code/rmhmc.py does, in C instead of NumPy) on a smaller sample, for an apples-to-apples number
    if literal_budget_s > 0 and D <= 64:
        rs = np.random.RandomState(0)
        w = np.full((cores, D), 1e-3); p = rs.randn(cores, D)
        with oracle.context(M, D, cores, flags=flags | _capi.FLAG_ORACLE_LITERAL) as ctx:
            ctx.set_data(XX, t)
            t0 = time.perf_counter(); ctx.leapfrog(w, p, eps, 1, 1, K); dtl = time.perf_counter() - t0
        out["reference_algorithm"] = {"value": cores / dtl, "unit": "leapfrog-steps/s", "cores": cores, "kind": "port",
                                      "stands_in_for": "the reference NumPy path (BASELINE.md 3.1): same O(M D^3) algorithm compiled from C, "
                                                       "so it is FASTER than rmhmc.py's interpreter loop and the GPU/CPU ratio it gives is conservative",
                                      "sample": "%d chains x 1 leapfrog step incl. the set-up block (rmhmc.py:50-77), literal "
                                                "tensor-forming C restatement of rmhmc.py, %.1f s" % (cores, dtl)}
    # ... and the same literal algorithm in NumPy (oracle/rmhmc_numpy.py, pinned to the golden tapes like the C oracle): what north_star calls
    # "the reference NumPy path".  One chain, as the reference runs; NumPy's BLAS uses the host cores it finds.
    if literal_budget_s > 0 and D <= 64:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import rmhmc_numpy as rn
        tt = np.ascontiguousarray(t, dtype=np.float64).ravel()
        pt = rn.Point(XX, tt, np.full(D, 1e-3))                      # the set-up block (untimed: a step's own block is timed below)
        pn = np.random.RandomState(0).randn(D)
        t0 = time.perf_counter(); nst = 0
        while nst < 1 or (time.perf_counter() - t0 < literal_budget_s / 3 and nst < 50):
            pt, pn = rn.leapfrog(XX, tt, pt, pn, eps, 1.0, K, guards=False)
            nst += 1
        dtn = time.perf_counter() - t0
        out["reference_numpy"] = {"value": nst / dtn, "unit": "leapfrog-steps/s", "cores": cores, "kind": "port",
                                  "sample": "1 chain x %d leapfrog steps, NumPy restatement of rmhmc.py's literal O(M D^3) algorithm "
                                            "(einsum tensor, LU inv / solve), %.1f s" % (nst, dtn)}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--chains", type=int, default=0, help="chains per GPU (default: the workload's)")
    ap.add_argument("--i8-slices", type=int, default=-1, help="4..7: assemble the metric (and the leverages) on the int8 matrix cores from that "
                    "many exact byte slices per operand (RMHMC_FLAG_INT8_METRIC, 8 < D <= 256); 0: fp64 matrix cores; -1 (default): 6 slices "
                    "(error of G 2e-14, the level of fp64 summation) where the path applies and the batch fills its 128-chain tiles")
    ap.add_argument("--no-alternates", action="store_true", help="skip the short extra runs with the other metric-assembly variants")
    ap.add_argument("--compat", type=int, default=0, help="1: reference-compatible momentum (L'z) and guards; 0: corrected (default, see DESIGN.md)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--ess-iters", type=int, default=-1,
                    help="run RMHMC for this many post-burn-in transitions (+100 burn-in) per chain and report min-ESS/sec, the second half of "
                         "the metric; -1 (default): 200 (c1-c3; config 5 at 1 s per global step: 0 = off), 0: off")
    ap.add_argument("--burn-in-steps", type=int, default=-1,
                    help="untimed global leapfrog steps between chains_init and the warmup, so that the timed steps run on chains at "
                         "stationarity (the metric's window is the post-burn-in sampling phase, rmhmc.py:194-198).  Default: 300 for the "
                         "large batched workloads (about 85 transitions; 80 at D > 64), 0 otherwise; alternates.cold_start times the first steps from theta0")
    ap.add_argument("--no-fp64-roofline", action="store_true", help="skip the extra fp64-matrix-core run behind roofline_fp64")
    ap.add_argument("--no-graph", action="store_true", help="plain launches instead of hipGraph replay (library option graph = 0)")
    ap.add_argument("--option", action="append", default=[], metavar="KEY=VALUE",
                    help="a tuning option of include/rmhmc.h (rmhmc_create_opts / rmhmc_set_option), repeatable; the active set is reported in config.options")
    ap.add_argument("--save-state", default="", help="after the burn-in steps, write the chain state (rmhmc_chains_state: a complete checkpoint) to this .npz and exit")
    ap.add_argument("--load-state", default="", help="start from a checkpoint written by --save-state instead of running the burn-in steps (bit-exact "
                    "resume, rmhmc_chains_restore): lets a profiler see stationary steps only (tools/profile.sh)")
    args = ap.parse_args()
    opts = {}
    for kv in args.option:
        k, _, v = kv.partition("=")
        opts[k] = int(v)
    if args.no_graph:
        opts["graph"] = 0

    import torch
    import torch.distributed as dist
    from riemannhamiltonianmontecarlo_amd import _capi

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("BENCH_BACKEND", "nccl")   # "gloo": rehearsal of the multi-rank logic on fewer GPUs than ranks
    ndev = torch.cuda.device_count()
    dev = local_rank % max(ndev, 1)
    ddev = torch.device("cuda", dev) if backend == "nccl" else torch.device("cpu")
    force_dist = os.environ.get("BENCH_FORCE_DIST") == "1" and "RANK" in os.environ   # rehearse the RCCL calls with one rank
    if world > 1 or force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(dev)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group(backend)
    if args.gpus != world and rank == 0 and world > 1:
        print("warning: --gpus %d but WORLD_SIZE %d" % (args.gpus, world), file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP library has no CPU fallback")

    wl = WORKLOADS[args.workload]
    n = args.chains or wl["chains"]
    XX, t = load_problem(args.workload)
    M, D = XX.shape
    L, eps, K = 6, 0.5, 4  # reference defaults, rmhmc.py:13
    flags = _capi.COMPAT if args.compat else 0
    auto_i8 = args.i8_slices < 0
    if auto_i8:
        args.i8_slices = 6 if (8 < D <= 256 and n * float(M) * D * D >= 1e9) else 0
    gpu_flags = flags | (_capi.int8_metric_flags(args.i8_slices) if args.i8_slices else 0)
    if auto_i8 and args.i8_slices:
        gpu_flags |= _capi.FLAG_INT8_CERTIFY   # as the shims do: rmhmc_set_data checks the fixed-point error bound for this data

    lib = _capi.load_hip_library()  # raises if the extension is not built

    def make_context(fl):
        """(rmhmc_create_opts takes every key; the run-time ones could also be changed later with rmhmc_set_option)"""
        c = lib.context(M, D, n, flags=fl, device=dev, options=opts or None)
        c.set_data(XX, t)
        return c

    ctx = make_context(gpu_flags)
    active_options = ctx.options()
    i8_bound, i8_active = ctx.int8_certificate()
    if args.i8_slices and not i8_active:   # not certified to 1e-9: the library runs this data on the fp64 matrix cores
        args.i8_slices = 0
    if args.burn_in_steps < 0:
        big_batch = 8 < D <= 256 and n * float(M) * D * D >= 1e9
        args.burn_in_steps = (300 if D <= 64 else 80) if big_batch else 0   # (config 5: 0.5 s per global step)
    if args.load_state:
        ck = np.load(args.load_state)
        assert ck["w"].shape == (n, D) and int(ck["burn_in_steps"]) == args.burn_in_steps, "checkpoint of another run"
        ctx.chains_init(theta0=ck["w"], seed=2024, chain_offset=rank * n, L=L, eps=eps, K=K)
        ctx.chains_restore(ck["iters"], ck["accepted"])
    else:
        ctx.chains_init(seed=2024, chain_offset=rank * n, L=L, eps=eps, K=K)
        if args.burn_in_steps:
            ctx.chains_run(args.burn_in_steps)   # untimed: from theta0 = 1e-3 (rmhmc.py:27) to stationarity
    if args.save_state:
        w_ck, it_ck, acc_ck = ctx.chains_state()
        np.savez(args.save_state, w=w_ck, iters=it_ck, accepted=acc_ck, burn_in_steps=np.int64(args.burn_in_steps))
        ctx.close()
        return

    def barrier():
        torch.cuda.synchronize()
        if world > 1 or force_dist:
            dist.barrier()
        torch.cuda.synchronize()

    ctx.chains_run(args.warmup)
    barrier()
    t0 = time.perf_counter()
    ctx.chains_run(args.steps)   # synchronous: returns when the device is idle.  Product path: graph replay, no events
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    barrier()
    elapsed = t1 - t0
    if world > 1 or force_dist:
        tt = torch.tensor([elapsed], device=ddev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    KT_NAMES = ("assemble", "assemble_i8", "assemble_i8_inner", "assemble_i8_inner_delta", "assemble_i8_delta", "vsplit", "leverage", "leverage_i8", "qsplit", "trvec", "rowpass", "mompass", "factor", "small",
                "fused", "medium", "total")
    # un-timed repetition of the same K steps with HIP events around every launch (on the library's own stream)
    ctx.kernel_time("enable"); ctx.kernel_time("reset")
    tk0 = time.perf_counter(); ctx.chains_run(args.steps); tk1 = time.perf_counter()
    kt = {k: ctx.kernel_time(k) for k in KT_NAMES}
    ctx.kernel_time("disable")
    finite = True
    if world > 1 or force_dist:
        # the one exchange of the sharded path: the chain positions are gathered at write-out straight from HBM
        # (rmhmc_chains_state_dev writes into a torch tensor on this GPU; RCCL sends that buffer over xGMI)
        wt, it_t, acc_t = ctx.chains_state_dev(torch.device("cuda", dev))
        if backend != "nccl":
            wt = wt.to(ddev)
        gathered = [torch.empty_like(wt) for _ in range(world)] if rank == 0 else None
        dist.gather(wt, gathered, dst=0)
        if rank == 0:
            finite = all(bool(torch.isfinite(g).all()) for g in gathered)
            gathered_chains = int(sum(g.shape[0] for g in gathered))
        iters, acc = it_t.cpu().numpy(), acc_t.cpu().numpy()
        finite = finite and bool(torch.isfinite(wt).all())
    else:
        w_end, iters, acc = ctx.chains_state()
        finite = bool(np.isfinite(w_end).all())
        gathered_chains = n
    acc_rate = float(acc.sum()) / max(1.0, float(iters.sum()))

    ess = None
    if args.ess_iters < 0:
        args.ess_iters = 0 if args.workload == "c5" else 200
    if args.ess_iters > 0:
        # second half of the metric: sum over chains of (min over dims of the per-chain ESS) / seconds of the
        # post-burn-in phase (TimeTaken semantics, rmhmc.py:194-198).  ESS by tools.CalculateESS semantics with
        # the MATLAB FFT length (no wrap-around); estimated on a subset of chains to bound the host FFT work.
        burn = 100
        st = ctx.sample_stats_dev(torch.device("cuda", dev), burn + args.ess_iters, burn, L=L, eps=eps, K=K, seed=7, chain_offset=rank * n)
        me = torch.nan_to_num(st["ess"], nan=float("inf")).amin(dim=1)        # per chain: min over dimensions (NaN: a constant coordinate)
        me = me[torch.isfinite(me)]                                            # (a chain with no finite ESS at all counts for nothing)
        tot = float(me.sum()); secs = st["seconds"]; lsteps = float(st["leapfrog_steps"].sum()); acc_n = float(st["accepted"].sum())
        if world > 1:
            tt = torch.tensor([tot, lsteps, acc_n], device=ddev, dtype=torch.float64); dist.all_reduce(tt)
            ts = torch.tensor([secs], device=ddev, dtype=torch.float64); dist.all_reduce(ts, op=dist.ReduceOp.MAX)
            tot, lsteps, acc_n, secs = float(tt[0]), float(tt[1]), float(tt[2]), float(ts[0])
        ess = {"min_ess_per_sec": tot / secs, "unit": "min-ESS/s (sum over chains of min_d ESS_d, / TimeTaken)", "seconds": secs,
               "post_burn_in_transitions": args.ess_iters, "burn_in": burn, "compat": bool(args.compat),
               "mean_min_ess_per_chain": tot / (n * world), "chains": n * world,
               "leapfrog_steps_per_sec_during_sampling": lsteps / secs,
               "acceptance": acc_n / float((burn + args.ess_iters) * n * world), "graph_replay": bool(active_options.get("graph", 1)),
               "note": "per-chain ESS (MATLAB CalculateStatistics.m semantics: ESS per chain, min over dimensions; tools.py:32-74 "
                       "estimator with linear autocovariances = the MATLAB FFT length, no wrap-around), summed over all chains; seconds = "
                       "the post-burn-in window of rmhmc.py:194-198; computed on the device by rmhmc_sample_stats_dev (no sample transfer)"}

    if rank == 0:
        total_steps = world * n * args.steps
        value = total_steps / elapsed
        bytes_step = 80.0 * M * D          # (2K+2) passes over X per chain per leapfrog step, K=4
        flops_step = 6.0 * M * D * D + 40.0 * M * D + 2.0 * D ** 3
        a_s, a_n = kt["assemble"]
        pass_bytes = 8.0 * M * D * n       # one assembly launch = one pass over X for every chain on this GPU
        roof = None
        f_s, f_n = kt["fused"]
        if f_n > 0:
            # small-problem path: one kernel does everything; a launch covers `steps` leapfrog steps of every chain
            f_avg = f_s / f_n
            units = n * args.steps / f_n
            achieved = units * bytes_step / f_avg
            roof = {"bound": "hbm", "kernel": "k_fused_small (whole leapfrog step, X resident in LDS, fp64 VALU)",
                    "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": achieved / HBM_PEAK, "traffic": None,
                    "avg_launch_ms": f_avg * 1e3, "launches": f_n,
                    "fp64_tflops": units * flops_step / f_avg / 1e12, "fp64_frac": units * flops_step / f_avg / FP64_MFMA_PEAK,
                    "note": "achieved = algorithmic bytes (80*M*D per chain per leapfrog step, SURVEY 8(d)) / measured launch time; "
                            "X (64 KB at config 2) is loaded into LDS once per launch, so real HBM traffic is ~0 and the kernel is "
                            "bound by fp64 VALU issue and exp/log latency, not by HBM"}
        elif kt["medium"][1] > 0:
            # small-batch path: one launch = one leapfrog step of every chain (one workgroup per chain)
            m_s, m_n = kt["medium"]
            m_avg = m_s / m_n
            achieved = n * bytes_step / m_avg
            roof = {"bound": "hbm", "kernel": "k_step_medium (whole leapfrog step in one launch, one 256-thread workgroup per chain)",
                    "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": achieved / HBM_PEAK, "traffic": None,
                    "avg_launch_ms": m_avg * 1e3, "launches": m_n,
                    "fp64_tflops": n * flops_step / m_avg / 1e12,
                    "note": "achieved = algorithmic bytes (80*M*D per chain per leapfrog step, SURVEY 8(d)) / measured launch time.  With one "
                            "chain this path is bound by the serial dependency chain of the step (about 70 us of barriers, LDS round trips "
                            "and the wave-0 factorisations) plus two bookkeeping launches, not by any throughput limit"}
        elif kt["assemble_i8"][1] > 0:
            # int8 metric path: the assembly is a sliced integer GEMM, S(S+1)/2 byte products per fp64 product
            S = args.i8_slices
            i_s, i_n = kt["assemble_i8"]
            i_avg = i_s / i_n
            NP = D * (D + 1) // 2
            ops = 2.0 * n * M * NP * (S * (S + 1) // 2)
            roof = {"bound": "mfma", "kernel": "k_assemble_i8 (sum_n v_n x_na x_nb as a sliced int8 GEMM, %d slices, v_mfma_i32_32x32x32_i8)" % S,
                    "achieved": ops / i_avg / 1e12, "peak": INT8_MFMA_PEAK / 1e12, "unit": "TOP/s", "frac": ops / i_avg / INT8_MFMA_PEAK,
                    "traffic": None, "avg_launch_ms": i_avg * 1e3, "launches": i_n,
                    "measured_mfma_stream_peak": INT8_MFMA_STREAM / 1e12,
                    "frac_of_measured_mfma_stream": ops / i_avg / INT8_MFMA_STREAM,
                    "feed_free_rate_on_real_bytes": INT8_FEED_FREE_REAL / 1e12,
                    "feed_free_note": "profiles/r03_i8_real_ceiling.txt: the same tile on the same bytes without LDS-DMA feed and fragment re-reads "
                                      "sustains 2.8-2.9 POP/s (the chip gives the clock back under load); the product reached 0.83-0.87 of that on the same box",
                    "fp64_equivalent_tflops": 2.0 * n * M * NP / i_avg / 1e12,
                    "fp64_equivalent_frac_of_fp64_mfma_peak": 2.0 * n * M * NP / i_avg / FP64_MFMA_PEAK,
                    "vsplit_avg_launch_ms": kt["vsplit"][0] / max(1, kt["vsplit"][1]) * 1e3,
                    # the assemblies of the position fixed-point iterates before the last: the 5 most significant slices (15 products)
                    "inner_iterate_assembly": ({"slices": S - 1, "launches": kt["assemble_i8_inner"][1],
                                                "avg_launch_ms": kt["assemble_i8_inner"][0] / kt["assemble_i8_inner"][1] * 1e3,
                                                "frac": 2.0 * n * M * NP * ((S - 1) * S // 2) / (kt["assemble_i8_inner"][0] / kt["assemble_i8_inner"][1]) / INT8_MFMA_PEAK}
                                               if kt["assemble_i8_inner"][1] else None),
                    # the assembly of the evaluation that ends a step: G(last iterate) + the GEMM of the v differences, from as many
                    # slices as the largest difference needs (4 at stationarity: 10 slice products; 5 / 6 in the first steps from theta0)
                    "delta_assembly": ({"launches": kt["assemble_i8_delta"][1],
                                        "avg_launch_ms": kt["assemble_i8_delta"][0] / kt["assemble_i8_delta"][1] * 1e3}
                                       if kt["assemble_i8_delta"][1] else None),
                    "survey8d_streaming_model_frac_NOT_A_BOUND": (value / world) * bytes_step / HBM_PEAK,
                    "step_fp64_frac": (value / world) * flops_step / FP64_MFMA_PEAK,
                    "note": "survey8d_streaming_model_frac = steps/s x 80 M D bytes / 8 TB/s, the figure SURVEY 8(d) / north_star ask for; it can "
                            "exceed 1 because X is shared by all chains (cache resident), so it is a model, not a physical bound.  "
                            "achieved = int8 multiply-adds issued for the unpadded problem (chains x D(D+1)/2 pairs x M rows x S(S+1)/2 "
                            "slice products) / measured launch time; fp64_equivalent = the fp64 flops of the same assembly"}
        elif a_n > 0:
            a_avg = a_s / a_n
            achieved = pass_bytes / a_avg
            if D > 64:  # large-D path: one launch = all block pairs; compute bound (AI above the ridge)
                nbk = (D + 63) // 64
                tiles = nbk * 10 + (nbk * (nbk + 1) // 2 - nbk) * 16
                fl = n * (M / 4.0) * tiles * 2048.0
                roof = {"bound": "mfma", "kernel": "k_assemble_pair (X' diag(v) X, one wave per (chain, 64-column block pair), fp64 MFMA)",
                        "achieved": fl / a_avg / 1e12, "peak": FP64_MFMA_PEAK / 1e12, "unit": "TFLOP/s", "frac": fl / a_avg / FP64_MFMA_PEAK,
                        "traffic": None, "avg_launch_ms": a_avg * 1e3, "launches": a_n,
                        "alg_hbm_gbs": achieved / 1e9, "survey8d_streaming_model_frac_NOT_A_BOUND": (value / world) * bytes_step / HBM_PEAK,
                        "step_fp64_frac": (value / world) * flops_step / FP64_MFMA_PEAK,
                        "note": "achieved = MFMA flops issued by one assembly launch (all block pairs) / measured launch time"}
            else:
                nb16 = (D + 15) // 16
                fl = n * (M / 4.0) * (nb16 * (nb16 + 1) // 2) * 2048.0
                roof = {"bound": "hbm", "kernel": "k_assemble (X' diag(v) X on fp64 MFMA)", "achieved": achieved / 1e9,
                        "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": achieved / HBM_PEAK, "traffic": None,
                        "avg_launch_ms": a_avg * 1e3, "launches": a_n,
                        "mfma_tflops": fl / a_avg / 1e12, "mfma_frac": fl / a_avg / FP64_MFMA_PEAK,
                        "survey8d_streaming_model_frac_NOT_A_BOUND": (value / world) * bytes_step / HBM_PEAK,
                        "step_fp64_frac": (value / world) * flops_step / FP64_MFMA_PEAK,
                        "note": "achieved = algorithmic bytes (8*M*D per chain per pass) / measured launch time; X is shared by all "
                                "chains and cache resident, so DRAM traffic is far below the algorithmic bytes (see DESIGN.md)"}
            pmc = os.path.join(ROOT, "profiles", "traffic_%s.json" % args.workload)
            if os.path.exists(pmc):
                roof["traffic"] = json.load(open(pmc)).get("assemble_bytes_per_launch")
        if roof is not None and kt["assemble_i8"][1] > 0:
            pmc = os.path.join(ROOT, "profiles", "traffic_%s.json" % args.workload)
            if os.path.exists(pmc):
                roof["traffic"] = json.load(open(pmc)).get("assemble_i8_x%d_bytes_per_launch" % args.i8_slices)
        out = {
            "metric": "leapfrog-steps/sec (whole node)", "value": value, "unit": "leapfrog-steps/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64 (the two O(M D^2) contractions as exact int8-sliced integer GEMMs)" if args.i8_slices else "f64",
            "dtype_detail": ("f64 throughout; the two O(M D^2) contractions (metric assembly, leverages) as exact integer GEMMs: operands cut into %d "
                             "signed-byte slices, int8 MFMA with int32 accumulation, combined in f64 (G to ~2e-14 of the f64 oracle at 6 slices)%s"
                             % (args.i8_slices, "; the metric of the position fixed-point iterates before the last (it only steers the next iterate) "
                                "from the 5 most significant slices (effect on theta after a step < 1e-11); the metric at the end of a step and that of the second position iterate as the "
                                "predecessor's G + the exact integer GEMM of the v differences (4-6 slices as their size needs, identical grids)" if args.i8_slices == 6 else ""))
                            if args.i8_slices else "f64 throughout (fp64 matrix cores)",
            "data": "synthetic" if args.workload != "c1" else "bundled australian.csv",
            "config": {"workload": "%s: %s" % (args.workload, wl["desc"]), "chains_per_gpu": n, "chains_total": n * world,
                       "D": D, "M": M, "leapfrog_L": L, "step_size": eps, "fixed_point_K": K,
                       "compat": bool(args.compat), "options": active_options,
                       "chain_state": ("stationary: %d untimed global steps from theta0 before the warmup (the metric's window is the post-burn-in "
                                       "sampling phase, rmhmc.py:194-198); alternates.cold_start = the first steps from theta0" % args.burn_in_steps)
                       if args.burn_in_steps else "first steps from theta0 = 1e-3 (rmhmc.py:27)",
                       "metric_assembly": (("int8 x %d slices" % args.i8_slices) + (" (position fixed-point iterates before the last: 5; "
                                           "RMHMC_FLAG_INT8_INNER_FULL = 6 everywhere, see alternates.int8_x6_inner_full)" if args.i8_slices == 6 else ""))
                       if args.i8_slices else "fp64",
                       "int8_error_certificate": i8_bound if i8_bound > 0 else None, "parallelism": "chains sharded over %d GPU(s), no data-path collective" % world},
            "roofline": roof,
            "kernel_seconds": {k: {"seconds": v[0], "launches": v[1]} for k, v in kt.items()},
            "kernel_seconds_from": "un-timed repetition of the same %d steps with HIP events around every launch (%.1f ms per step there; the "
                                   "timed region itself runs the hipGraph path without events)" % (args.steps, (tk1 - tk0) / args.steps * 1e3),
            "all_finite": finite, "acceptance_rate": acc_rate, "gathered_chains": gathered_chains,
        }
        if roof is not None and roof.get("traffic") is not None:
            roof["traffic_source"] = "profiles/traffic_%s.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command (per launch)" % args.workload
        if ess is not None:
            out["min_ess"] = ess
            out["min_ess_per_sec"] = ess["min_ess_per_sec"]
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(XX, t, flags, L, eps, K)
    ctx.close()
    if rank == 0:
        big = 8 < D <= 256 and n * float(M) * D * D >= 1e9

        def run_variant(sl, steps, timing, extra=0, burn=None):
            c2 = make_context(flags | extra | (_capi.int8_metric_flags(sl) if sl else 0))
            c2.chains_init(seed=2024, chain_offset=rank * n, L=L, eps=eps, K=K)
            burn = args.burn_in_steps if burn is None else burn
            if burn:
                c2.chains_run(burn)
            c2.chains_run(min(args.warmup, 2) or 1)
            torch.cuda.synchronize(); ta = time.perf_counter()
            c2.chains_run(steps)
            torch.cuda.synchronize(); tb = time.perf_counter()
            k2 = None
            if timing:
                c2.kernel_time("enable"); c2.kernel_time("reset")
                c2.chains_run(steps)
                k2 = {k: c2.kernel_time(k) for k in KT_NAMES}
            c2.close()
            return {"value": n * steps / (tb - ta), "ms_per_step": (tb - ta) / steps * 1e3}, k2

        if world == 1 and big and args.i8_slices and not args.no_fp64_roofline and D <= 64:
            # the same workload for the same number of steps with the metric on the fp64 matrix cores: the number in the SAME arithmetic
            # as the reference, with its own roofline (k_assemble: MFMA flops issued / launch time / fp64 MFMA peak)
            res, k2 = run_variant(0, args.steps, True)
            a_s2, a_n2 = k2["assemble"]
            nb16 = (D + 15) // 16
            fl = n * (M / 4.0) * (nb16 * (nb16 + 1) // 2) * 2048.0
            a_avg2 = a_s2 / max(1, a_n2)
            out["roofline_fp64"] = {"value": res["value"], "unit": "leapfrog-steps/s", "ms_per_step": res["ms_per_step"], "steps": args.steps,
                                    "bound": "mfma", "kernel": "k_assemble (X' diag(v) X on v_mfma_f64_16x16x4_f64, lower-triangle tiles only)",
                                    "achieved": fl / a_avg2 / 1e12, "peak": FP64_MFMA_PEAK / 1e12, "unit_roofline": "TFLOP/s",
                                    "frac": fl / a_avg2 / FP64_MFMA_PEAK, "avg_launch_ms": a_avg2 * 1e3, "launches": a_n2,
                                    "kernel_seconds": {k: {"seconds": v[0], "launches": v[1]} for k, v in k2.items() if v[1]},
                                    "survey8d_streaming_model_frac_NOT_A_BOUND": res["value"] * bytes_step / HBM_PEAK,
                                    "note": "dtype f64 in every kernel (no integer slicing); achieved = MFMA flops issued by one assembly launch "
                                            "(chains x M/4 x 10 tiles x 2048) / its average launch time in an un-timed repetition with events"}
        if world == 1 and not args.no_alternates and big:
            # the same workload with the other metric-assembly variants, 3 steps each (not the headline)
            alts = {}
            for name, sl in (("fp64_mfma", 0), ("int8_x5", 5), ("int8_x6", 6)):
                if sl == args.i8_slices or (sl == 0 and "roofline_fp64" in out):
                    continue
                alts[name], _ = run_variant(sl, 3, False)
            if args.i8_slices == 6:   # all four assemblies of a step from 6 slices
                alts["int8_x6_inner_full"], _ = run_variant(6, 3, False, _capi.FLAG_INT8_INNER_FULL)
            if args.burn_in_steps and args.i8_slices:   # the headline's arithmetic on the first steps from theta0 (no burn-in)
                alts["cold_start"], _ = run_variant(args.i8_slices, args.steps, False, burn=0)
            if "roofline_fp64" in out:
                alts["fp64_mfma"] = {"value": out["roofline_fp64"]["value"], "ms_per_step": out["roofline_fp64"]["ms_per_step"]}
            out["alternates"] = alts
        print(json.dumps(out))
    if world > 1 or force_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

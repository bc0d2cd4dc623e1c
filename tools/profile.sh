#!/bin/bash
# Collect the rocprofv3 evidence for bench.py on the GPU box (run through gpurun from the repo root):
#   tools/profile.sh TAG [WORKLOAD] [STEPS] [BURN]
#   0. un-profiled: BURN global steps from theta0, chain state saved as a checkpoint (rmhmc_chains_state is a complete one)
#   1. kernel trace + stats of STEPS global steps from that checkpoint                  -> gpurun_out/prof_<tag>/stats
#   2. PMC passes, one counter group per run (separate runs, as MI355X_MICROARCH.md prescribes; never together with a trace), ONE
#      timed step from the same checkpoint each:
#      FETCH_SIZE | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE | SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT
# So every duration, cycle count and byte count is of a launch on chains at stationarity (resume is bit exact).  All profiled runs pass
# --no-graph: hipGraph replay under rocprofv3's queue interceptor faults in librocprofiler-sdk 7.2 (tools/rocprof_queue_repro.hip, DESIGN
# section 6); the kernels and their order are the same, the launches just reach the queue one by one.
# Summaries are written by tools/summarize_profile.py into gpurun_out/prof_<tag>/ and copied to profiles/ (committed).
set -o pipefail
TAG=${1:-r03}
WL=${2:-c3}
STEPS=${3:-6}
BURN=${4:--1}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
CK=/tmp/rmhmc_ck_$TAG.npz
BENCH0="python3 bench.py --workload $WL --no-cpu-baseline --no-alternates --no-fp64-roofline --ess-iters 0 --burn-in-steps $BURN"
$BENCH0 --save-state $CK || exit 1
BENCH="$BENCH0 --load-state $CK --no-graph"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $BENCH --steps $STEPS --warmup 1 > $OUT/bench_stats.json 2> $OUT/stats.err || { tail -5 $OUT/stats.err; exit 1; }
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $BENCH --steps 1 --warmup 0 > $OUT/bench_fetch.json 2> $OUT/fetch.err || { tail -5 $OUT/fetch.err; exit 1; }
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $BENCH --steps 1 --warmup 0 > $OUT/bench_write.json 2> $OUT/write.err || { tail -5 $OUT/write.err; exit 1; }
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/mfma -- $BENCH --steps 1 --warmup 0 > $OUT/bench_mfma.json 2> $OUT/mfma.err || { tail -5 $OUT/mfma.err; exit 1; }
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/lds -- $BENCH --steps 1 --warmup 0 > $OUT/bench_lds.json 2> $OUT/lds.err || { tail -5 $OUT/lds.err; exit 1; }
python3 tools/summarize_profile.py $OUT $TAG $WL
# (the raw traces are large: only the summaries travel back)
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -size +8M -delete

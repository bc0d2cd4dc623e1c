#!/bin/bash
# Collect the rocprofv3 evidence for bench.py on the GPU box (run through gpurun from the repo root):
#   1. kernel trace + stats (per-kernel time)      -> gpurun_out/prof_<tag>/stats
#   2. PMC passes, one counter group per run (separate runs, as MI355X_MICROARCH.md prescribes; never together with a trace):
#      FETCH_SIZE | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE | SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT
# Summaries are written by tools/summarize_profile.py into gpurun_out/prof_<tag>/ and copied to profiles/ (committed).
set -o pipefail
TAG=${1:-r02}
WL=${2:-c3}
STEPS=${3:-3}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
# (--burn-in-steps 150: past the first trajectories from theta0, where the delta assembly still needs 5 or 6 slices, and few enough
#  dispatches for the trace; the counter passes, ~50 ms per dispatch, take the very first step from theta0, whose delta assembly
#  also runs on four slices)
BURN=${4:-150}
BENCH0="python3 bench.py --workload $WL --no-cpu-baseline --no-alternates --no-fp64-roofline --ess-iters 0"
BENCH="$BENCH0 --burn-in-steps $BURN"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $BENCH --steps $STEPS --warmup 1 > $OUT/bench_stats.json 2> $OUT/stats.err || { tail -5 $OUT/stats.err; exit 1; }
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $BENCH0 --burn-in-steps 0 --steps 1 --warmup 0 > $OUT/bench_fetch.json 2> $OUT/fetch.err || { tail -5 $OUT/fetch.err; exit 1; }
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $BENCH0 --burn-in-steps 0 --steps 1 --warmup 0 > $OUT/bench_write.json 2> $OUT/write.err || { tail -5 $OUT/write.err; exit 1; }
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/mfma -- $BENCH0 --burn-in-steps 0 --steps 1 --warmup 0 > $OUT/bench_mfma.json 2> $OUT/mfma.err || { tail -5 $OUT/mfma.err; exit 1; }
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/lds -- $BENCH0 --burn-in-steps 0 --steps 1 --warmup 0 > $OUT/bench_lds.json 2> $OUT/lds.err || { tail -5 $OUT/lds.err; exit 1; }
python3 tools/summarize_profile.py $OUT $TAG $WL

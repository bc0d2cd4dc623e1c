#!/bin/bash
# Collect the rocprofv3 evidence for bench.py on the GPU box (run through gpurun from the repo root):
#   1. kernel trace + stats (per-kernel time)      -> gpurun_out/prof_<tag>/stats
#   2. PMC FETCH_SIZE and 3. PMC WRITE_SIZE passes (separate runs, as MI355X_MICROARCH.md prescribes)
# Summaries are written by tools/summarize_profile.py into profiles/ (committed).
set -o pipefail
TAG=${1:-r01}
WL=${2:-c3}
STEPS=${3:-3}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --workload $WL --steps $STEPS --warmup 1 --no-cpu-baseline --no-alternates > $OUT/bench_stats.json 2> $OUT/stats.err || { tail -5 $OUT/stats.err; exit 1; }
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 bench.py --workload $WL --steps 1 --warmup 0 --no-cpu-baseline --no-alternates > $OUT/bench_fetch.json 2> $OUT/fetch.err || { tail -5 $OUT/fetch.err; exit 1; }
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 bench.py --workload $WL --steps 1 --warmup 0 --no-cpu-baseline --no-alternates > $OUT/bench_write.json 2> $OUT/write.err || { tail -5 $OUT/write.err; exit 1; }
python3 tools/summarize_profile.py $OUT $TAG $WL

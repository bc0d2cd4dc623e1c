#!/bin/bash
# tools/opt_ab.sh OUTDIR "OPT=A" "OPT=B" [reps]: the default bench with two settings of a library option, interleaved on one box
O=$1; A=$2; B=$3; R=${4:-3}; mkdir -p $O
BENCH="python bench.py --no-cpu-baseline --no-alternates --no-fp64-roofline --ess-iters 0 --steps 16 --warmup 2"
for rep in $(seq 1 $R); do
  for v in "$A" "$B"; do
    timeout -k 10 300 $BENCH --option $v > $O/${v//=/_}_$rep.json 2> $O/${v//=/_}_$rep.err || { echo "$v failed"; tail -3 $O/${v//=/_}_$rep.err; exit 1; }
    python - $O/${v//=/_}_$rep.json "$v" <<'P'
import json, sys
b = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("%-14s %9.0f steps/s  %.3f ms/step" % (sys.argv[2], b["value"], b["ms_per_step"]))
P
  done
done

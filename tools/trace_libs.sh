#!/bin/bash
# tools/trace_libs.sh OUTDIR REGEX LIB...: per-kernel durations (rocprofv3 --kernel-trace, 4 global steps from each library's own checkpoint,
# --no-graph) for a list of builds of the library (a path, or `product`), same box, same call.
set -o pipefail
O=$PWD/$1; RE="$2"; shift 2; mkdir -p $O; export TMPDIR=/tmp
B="python3 bench.py --workload c3 --no-cpu-baseline --no-alternates --no-fp64-roofline --ess-iters 0 --burn-in-steps 40"
for lib in "$@"; do
  v=$(basename $lib .so); if [ $lib = product ]; then unset RMHMC_HIP_LIB; else export RMHMC_HIP_LIB=$PWD/$lib; fi
  timeout -k 10 300 $B --save-state /tmp/ck_$v.npz > $O/save_$v.json 2> $O/save_$v.err || { tail -3 $O/save_$v.err; exit 1; }
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace_$v -- $B --load-state /tmp/ck_$v.npz --no-graph --steps 4 --warmup 1 > $O/t_$v.json 2> $O/t_$v.err || { tail -3 $O/t_$v.err; exit 1; }
  python3 - $O $v "$RE" <<'P'
import csv, glob, sys, collections, re
O, v, rx = sys.argv[1], sys.argv[2], re.compile(sys.argv[3])
t = glob.glob(O + "/trace_%s/**/*kernel_trace.csv" % v, recursive=True)[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(t)):
    n = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if rx.search(n): agg[n].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for n, vals in sorted(agg.items()):
    big = [x for x in vals if x > 0.2 * max(vals)]
    print("%-22s %-40s n=%3d avg=%8.1f us" % (v, n[-40:], len(big), sum(big) / len(big) / 1e3))
P
  find $O -name "*kernel_trace.csv" -delete
done

#!/bin/bash
# tools/fetch_ab.sh OUTDIR: HBM-side bytes (rocprofv3 --pmc FETCH_SIZE) and durations of the int8 GEMMs of one stationary step, product library
# against tools/librmhmc_hip_prev.so (tools/build_prev.sh), same box, same call.
set -o pipefail
O=$PWD/$1; mkdir -p $O; export TMPDIR=/tmp
B="python3 bench.py --workload c3 --no-cpu-baseline --no-alternates --no-fp64-roofline --ess-iters 0"
for v in new prev; do
  if [ $v = prev ]; then export RMHMC_HIP_LIB=$PWD/tools/librmhmc_hip_prev.so; else unset RMHMC_HIP_LIB; fi
  timeout -k 10 300 $B --save-state /tmp/ck_$v.npz > $O/save_$v.json 2> $O/save_$v.err || { tail -3 $O/save_$v.err; exit 1; }
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch_$v -- $B --load-state /tmp/ck_$v.npz --no-graph --steps 1 --warmup 0 > $O/f_$v.json 2> $O/f_$v.err || { tail -3 $O/f_$v.err; exit 1; }
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace_$v -- $B --load-state /tmp/ck_$v.npz --no-graph --steps 4 --warmup 1 > $O/t_$v.json 2> $O/t_$v.err || { tail -3 $O/t_$v.err; exit 1; }
  python3 - $O $v <<'P'
import csv, glob, sys, collections
O, v = sys.argv[1], sys.argv[2]
f = glob.glob(O + "/fetch_%s/**/*counter_collection.csv" % v, recursive=True)[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if "k_assemble_i8" in n and "tail" not in n or "k_leverage_i8<" in n:
        agg[n.split("(")[0][-40:]].append(float(r["Counter_Value"]))
for n, vals in sorted(agg.items()):
    big = [x for x in vals if x > 0.02 * max(vals)]
    print(v, "FETCH raw KiB", n, "n=%d avg=%.0f min=%.0f max=%.0f" % (len(big), sum(big) / len(big), min(big), max(big)))
t = glob.glob(O + "/trace_%s/**/*kernel_trace.csv" % v, recursive=True)[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(t)):
    n = r["Kernel_Name"]
    if "k_assemble_i8" in n and "tail" not in n or "k_leverage_i8<" in n:
        agg[n.split("(")[0][-40:]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for n, vals in sorted(agg.items()):
    big = [x for x in vals if x > 0.2 * max(vals)]
    print(v, "time us     ", n, "n=%d avg=%.1f" % (len(big), sum(big) / len(big) / 1e3))
P
  find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -size +8M -delete
done

#!/bin/bash
# tools/trace_ab.sh OUTDIR REGEX [bench args]: per-kernel durations (rocprofv3 --kernel-trace, 4 stationary global steps from a checkpoint,
# --no-graph) of the product library against tools/librmhmc_hip_prev.so (tools/build_prev.sh), same box, same call, two rounds each.
set -o pipefail
O=$PWD/$1; RE="$2"; shift 2; mkdir -p $O; export TMPDIR=/tmp
B="python3 bench.py --workload c3 --no-cpu-baseline --no-alternates --no-fp64-roofline --ess-iters 0 $@"
for rep in 1 2; do for v in prev new; do
  if [ $v = prev ]; then export RMHMC_HIP_LIB=$PWD/tools/librmhmc_hip_prev.so; else unset RMHMC_HIP_LIB; fi
  [ -f /tmp/ck_$v.npz ] || timeout -k 10 300 $B --save-state /tmp/ck_$v.npz > $O/save_$v.json 2> $O/save_$v.err || { tail -3 $O/save_$v.err; exit 1; }
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace_$v$rep -- $B --load-state /tmp/ck_$v.npz --no-graph --steps 4 --warmup 1 > $O/t_$v$rep.json 2> $O/t_$v$rep.err || { tail -3 $O/t_$v$rep.err; exit 1; }
  python3 - $O $v$rep "$RE" <<'P'
import csv, glob, sys, collections, re, json
O, v, rx = sys.argv[1], sys.argv[2], re.compile(sys.argv[3])
t = glob.glob(O + "/trace_%s/**/*kernel_trace.csv" % v, recursive=True)[0]
rows = sorted(csv.DictReader(open(t)), key=lambda r: int(r["Dispatch_Id"]))
first = next((i for i, r in enumerate(rows) if "k_iter_begin" in r["Kernel_Name"]), 0)
rows = rows[first:]
steps = max(1, sum(1 for r in rows if "k_iter_begin" in r["Kernel_Name"]))
agg = collections.defaultdict(list); tot = 0
for r in rows:
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); tot += d
    n = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if rx.search(n): agg[n].append(d)
try: val = json.loads(open(O + "/t_%s.json" % v).read().strip().splitlines()[-1])["ms_per_step"]
except Exception: val = float("nan")
print("%-6s kernel time per step %.3f ms (bench line %.3f ms/step)" % (v, tot / steps / 1e6, val))
for n, vals in sorted(agg.items()):
    big = [x for x in vals if x > 0.2 * max(vals)]
    print("%-6s %-44s n=%3d avg=%8.1f us  per step %.3f ms" % (v, n[-44:], len(big), sum(big) / len(big) / 1e3, sum(big) / steps / 1e6))
P
  find $O -name "*kernel_trace.csv" -delete
done; done

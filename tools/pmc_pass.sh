#!/bin/bash
# tools/pmc_pass.sh OUTDIR "COUNTER ..." [bench args]: one rocprofv3 --pmc pass (one timed global step of chains at stationarity, from a
# checkpoint, --no-graph) and the per-kernel average of every counter per launch.
set -o pipefail
O=$PWD/$1; C="$2"; shift 2; mkdir -p $O; export TMPDIR=/tmp
B="python3 bench.py --workload c3 --no-cpu-baseline --no-alternates --no-fp64-roofline --ess-iters 0 $@"
[ -f /tmp/ck_pmc.npz ] || timeout -k 10 300 $B --save-state /tmp/ck_pmc.npz > $O/save.json 2> $O/save.err || { tail -3 $O/save.err; exit 1; }
timeout -k 10 400 rocprofv3 --pmc $C --output-format csv -d $O/pmc -- $B --load-state /tmp/ck_pmc.npz --no-graph --steps 1 --warmup 0 > $O/b.json 2> $O/b.err || { tail -5 $O/b.err; exit 1; }
python3 - $O <<'P'
import csv, glob, sys, collections
O = sys.argv[1]
f = glob.glob(O + "/pmc/**/*counter_collection.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
first = next((i for i, r in enumerate(rows) if "k_iter_begin" in r["Kernel_Name"]), 0)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows[first:]:
    n = r["Kernel_Name"].split("(")[0].replace("void ", "")
    agg[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = sorted({c for d in agg.values() for c in d})
print("%-40s %5s " % ("kernel", "n") + " ".join("%22s" % c[-22:] for c in names))
for n, d in sorted(agg.items(), key=lambda kv: -sum(kv[1].get("SQ_WAVE_CYCLES", kv[1][names[0]]))):
    cnt = len(d[names[0]])
    print("%-40s %5d " % (n[-40:], cnt) + " ".join("%22.4g" % (sum(d[c]) / max(1, len(d[c]))) for c in names))
P
find $O -name "*counter_collection.csv" -size +8M -delete

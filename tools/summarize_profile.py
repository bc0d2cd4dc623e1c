#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel stats + PMC FETCH_SIZE/WRITE_SIZE passes) into small text/JSON
summaries under gpurun_out/ that are then copied into profiles/."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out, tag, wl = sys.argv[1], sys.argv[2], sys.argv[3]


def short(name):
    name = name.split("(")[0]
    for k in ("k_assemble", "k_leverage", "k_rowpass", "k_xtr", "k_factor_full", "k_factor_solve", "k_pos_first", "k_fused"):
        if k in name:
            return name[name.index(k):][:60]
    return name[:60]


lines = []
stats = glob.glob(os.path.join(out, "stats", "**", "*kernel_stats.csv"), recursive=True)
if stats:
    rows = list(csv.DictReader(open(stats[0])))
    lines.append("# rocprofv3 --kernel-trace --stats : per-kernel summary (%s, workload %s)" % (tag, wl))
    lines.append("%-62s %8s %14s %12s %7s" % ("kernel", "calls", "total_ms", "avg_us", "pct"))
    for r in rows[:25]:
        lines.append("%-62s %8s %14.3f %12.1f %7.2f" % (short(r["Name"]), r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                                                       float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
summary = {"tag": tag, "workload": wl}
for ctr in ("fetch", "write"):
    files = glob.glob(os.path.join(out, ctr, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        continue
    agg = defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(files[0])):
        k = short(r["Kernel_Name"])
        agg[k][0] += float(r["Counter_Value"]); agg[k][1] += 1
    cname = "FETCH_SIZE" if ctr == "fetch" else "WRITE_SIZE"
    lines.append("")
    lines.append("# rocprofv3 --pmc %s : per-kernel average per launch (raw counter, KiB units; FETCH_SIZE must be doubled on gfx950 for wide coalesced reads, MI355X_MICROARCH.md)" % cname)
    for k, (v, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:12]:
        lines.append("%-62s launches=%4d  avg_raw=%14.1f KiB" % (k, n, v / n))
        summary.setdefault(cname, {})[k] = {"launches": n, "avg_raw_kib": v / n}
bj = os.path.join(out, "bench_stats.json")
if os.path.exists(bj):
    try:
        lines.append(""); lines.append("# bench.py line of the same (profiled) run")
        lines.append(open(bj).read().strip())
    except Exception:
        pass
open(os.path.join(out, "summary_%s_%s.txt" % (tag, wl)), "w").write("\n".join(lines) + "\n")
json.dump(summary, open(os.path.join(out, "summary_%s_%s.json" % (tag, wl)), "w"), indent=1)
print("\n".join(lines))

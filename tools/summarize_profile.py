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
    for k in ("k_assemble", "k_leverage", "k_rowpass", "k_mompass", "k_trvec", "k_qsplit", "k_factor_full", "k_factor_solve", "k_pos_first", "k_fused"):
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
# matrix-pipe occupancy and effective clock per kernel (SQ cycles are quad-cycles; GRBM_GUI_ACTIVE sums the 8 XCDs), LDS activity
kernel_avg_ns = {}
if stats:
    for r in rows:
        kernel_avg_ns[short(r["Name"])] = float(r["AverageNs"])
for ctr, names in (("mfma", ("SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE")), ("lds", ("SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT"))):
    files = glob.glob(os.path.join(out, ctr, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        continue
    agg = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for r in csv.DictReader(open(files[0])):
        a = agg[short(r["Kernel_Name"])][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
    lines.append("")
    lines.append("# rocprofv3 --pmc %s : per-kernel average per launch" % " ".join(names))
    for k, d in sorted(agg.items(), key=lambda kv: -sum(v[0] for v in kv[1].values()))[:8]:
        vals = {n: d[n][0] / max(1, d[n][1]) for n in names if n in d}
        extra = ""
        if ctr == "mfma" and "GRBM_GUI_ACTIVE" in vals and k in kernel_avg_ns:
            clk = vals["GRBM_GUI_ACTIVE"] / 8.0 / (kernel_avg_ns[k] * 1e-9)          # Hz (un-profiled duration of the stats pass)
            cyc = vals["GRBM_GUI_ACTIVE"] / 8.0
            busy = vals.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024.0 * cyc) if cyc else 0.0   # 1024 SIMDs; as in profiles/r01_i8_gemm_probe.txt
            extra = "  clock %.2f GHz  MFMA busy %.1f %%" % (clk / 1e9, 100 * busy)
            vals["effective_clock_ghz"] = clk / 1e9; vals["mfma_busy_frac"] = busy
        lines.append("%-62s %s%s" % (k, "  ".join("%s=%.4g" % kv for kv in vals.items() if not kv[0].startswith(("eff", "mfma_b"))), extra))
        summary.setdefault(ctr, {})[k] = vals
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

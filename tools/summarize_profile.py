#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel trace + PMC passes) into small text/JSON summaries under gpurun_out/ that are then copied
into profiles/.  Every pass of tools/profile.sh starts from the SAME checkpoint of chains at stationarity (bench.py --load-state), so
durations, cycle counts and bytes belong to the same kind of launch; whatever a pass dispatches before its first global step (set-up,
the evaluation of the restored point) is dropped: only dispatches from the first k_iter_begin on are counted, and the number of
k_iter_begin dispatches is the number of global steps (launches per step = calls / steps)."""
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


# (Profiles up to the middle of round 3:) the delta assembly launched the S' = 6, 5, 4 instantiations of k_assemble_i8 / _tail / _tailsum and all but one returned at once
# (a few us, no memory traffic): those dispatches are counted apart, the averages are over the dispatches that did the work.
def is_gemm(name):
    return "k_assemble_i8" in name


def from_first_step(recs, key):
    """rows ordered by dispatch, from the first k_iter_begin (or one-launch step kernel) on; returns (rows, steps)"""
    recs = sorted(recs, key=lambda r: int(r[key]))
    first = next((i for i, r in enumerate(recs) if "k_iter_begin" in r["Kernel_Name"] or "k_step_medium" in r["Kernel_Name"]
                  or "k_fused_small" in r["Kernel_Name"]), 0)
    recs = recs[first:]
    ids = {r["Dispatch_Id"] for r in recs if "k_iter_begin" in r["Kernel_Name"]}
    return recs, max(1, len(ids))


lines = []
stats = glob.glob(os.path.join(out, "stats", "**", "*kernel_stats.csv"), recursive=True)
trace = glob.glob(os.path.join(out, "stats", "**", "*kernel_trace.csv"), recursive=True)
rows = []
if trace:
    per = defaultdict(list)
    trows, nsteps = from_first_step(list(csv.DictReader(open(trace[0]))), "Dispatch_Id")
    for r in trows:
        per[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    total = float(sum(sum(v) for v in per.values()))
    for name, d in per.items():
        idle = []
        if is_gemm(name):
            cut = 0.2 * max(d)
            idle = [x for x in d if x < cut]
            d = [x for x in d if x >= cut]
        rows.append({"Name": name, "Calls": len(d), "TotalDurationNs": float(sum(d)), "AverageNs": sum(d) / max(1, len(d)),
                     "Percentage": 100.0 * (sum(d) + sum(idle)) / total, "Returned": len(idle), "ReturnedNs": float(sum(idle))})
    rows.sort(key=lambda r: -r["TotalDurationNs"])
    lines.append("# rocprofv3 --kernel-trace --stats : per-kernel summary from the kernel trace (%s, workload %s)" % (tag, wl))
    lines.append("# (k_assemble_i8*: dispatches of the delta assembly that returned at once are listed as `returned`, not averaged in)")
    lines.append("# %d global steps of chains at stationarity (dispatches before the first k_iter_begin dropped); kernel time per step %.3f ms"
                 % (nsteps, total / nsteps / 1e6))
    lines.append("%-62s %8s %9s %14s %12s %7s" % ("kernel", "calls", "per_step", "total_ms", "avg_us", "pct"))
    for r in rows[:32]:
        lines.append("%-62s %8s %9.2f %14.3f %12.1f %7.2f%s" % (short(r["Name"]), r["Calls"], r["Calls"] / float(nsteps), r["TotalDurationNs"] / 1e6, r["AverageNs"] / 1e3, r["Percentage"],
                                                         ("   returned at once: %d (%.1f us each)" % (r["Returned"], r["ReturnedNs"] / r["Returned"] / 1e3)) if r["Returned"] else ""))
elif stats:
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
    recs, _ = from_first_step(list(csv.DictReader(open(files[0]))), "Dispatch_Id")
    peak = defaultdict(float)
    for r in recs:
        peak[r["Kernel_Name"]] = max(peak[r["Kernel_Name"]], float(r["Counter_Value"]))
    for r in recs:
        if is_gemm(r["Kernel_Name"]) and float(r["Counter_Value"]) < 0.02 * peak[r["Kernel_Name"]]:
            continue   # (a delta-assembly dispatch that returned at once)
        k = short(r["Kernel_Name"])
        agg[k][0] += float(r["Counter_Value"]); agg[k][1] += 1
    cname = "FETCH_SIZE" if ctr == "fetch" else "WRITE_SIZE"
    lines.append("")
    lines.append("# rocprofv3 --pmc %s : per-kernel average per launch (raw counter, KiB units; FETCH_SIZE must be doubled on gfx950 for wide coalesced reads, MI355X_MICROARCH.md)" % cname)
    for k, (v, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:16]:
        lines.append("%-62s launches=%4d  avg_raw=%14.1f KiB" % (k, n, v / n))
        summary.setdefault(cname, {})[k] = {"launches": n, "avg_raw_kib": v / n}
# matrix-pipe occupancy and effective clock per kernel (SQ cycles are quad-cycles; GRBM_GUI_ACTIVE sums the 8 XCDs), LDS activity
kernel_avg_ns = {}
if rows:
    for r in rows:
        kernel_avg_ns[short(r["Name"])] = float(r["AverageNs"])
for ctr, names in (("mfma", ("SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE")), ("lds", ("SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT"))):
    files = glob.glob(os.path.join(out, ctr, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        continue
    agg = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    recs, _ = from_first_step(list(csv.DictReader(open(files[0]))), "Dispatch_Id")
    peak = defaultdict(float)
    for r in recs:
        peak[(r["Kernel_Name"], r["Counter_Name"])] = max(peak[(r["Kernel_Name"], r["Counter_Name"])], float(r["Counter_Value"]))
    for r in recs:
        if is_gemm(r["Kernel_Name"]) and r["Counter_Name"] in ("GRBM_GUI_ACTIVE", "SQ_LDS_IDX_ACTIVE") \
                and float(r["Counter_Value"]) < 0.02 * peak[(r["Kernel_Name"], r["Counter_Name"])]:
            continue
        if is_gemm(r["Kernel_Name"]) and r["Counter_Name"] == "SQ_VALU_MFMA_BUSY_CYCLES" and float(r["Counter_Value"]) == 0.0:
            continue
        a = agg[short(r["Kernel_Name"])][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
    lines.append("")
    lines.append("# rocprofv3 --pmc %s : per-kernel average per launch" % " ".join(names))
    for k, d in sorted(agg.items(), key=lambda kv: -sum(v[0] for v in kv[1].values()))[:14]:
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
# roofline.traffic of bench.py: HBM-side bytes of one metric assembly (main launch + the tiles of the ragged pair block + their integer
# sum), FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for wide coalesced reads on gfx950, + WRITE_SIZE; KiB -> bytes
if "FETCH_SIZE" in summary and "WRITE_SIZE" in summary:
    tr = {"source": "profiles/%s_%s_pmc.json (tools/profile.sh %s: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes, chains at "
                    "stationarity from a checkpoint, dispatches from the first global step on)" % (tag, wl, tag),
          "correction": "FETCH_SIZE doubled (gfx950 reports half the bytes of wide coalesced reads, MI355X_MICROARCH.md); fabric-side counters, "
                        "Infinity-Cache hits included"}
    for S in (4, 5, 6, 7):
        f = w = 0.0
        for part in ("k_assemble_i8<%d, 4, 1>" % S, "k_assemble_i8<%d, 2, 1>" % S, "k_assemble_i8_tail<%d>" % S, "k_assemble_i8_tailsum<%d>" % S):
            f += summary["FETCH_SIZE"].get(part, {}).get("avg_raw_kib", 0.0)
            w += summary["WRITE_SIZE"].get(part, {}).get("avg_raw_kib", 0.0)
        if f > 0:
            tr["assemble_i8_x%d_fetch_raw_kib" % S] = f; tr["assemble_i8_x%d_write_raw_kib" % S] = w
            tr["assemble_i8_x%d_bytes_per_launch" % S] = (2.0 * f + w) * 1024.0
    # the one-launch delta assembly (whatever slice count its launches picked: 4 at stationarity)
    f = w = 0.0
    for part in ("k_assemble_i8_sel<4, 1>", "k_assemble_i8_tail_sel", "k_assemble_i8_tailsum_sel"):
        f += summary["FETCH_SIZE"].get(part, {}).get("avg_raw_kib", 0.0)
        w += summary["WRITE_SIZE"].get(part, {}).get("avg_raw_kib", 0.0)
    if f > 0:
        tr["assemble_i8_delta_fetch_raw_kib"] = f; tr["assemble_i8_delta_write_raw_kib"] = w
        tr["assemble_i8_delta_bytes_per_launch"] = (2.0 * f + w) * 1024.0
    for name in ("k_assemble<4>", "k_assemble<3>", "k_assemble<2>", "k_assemble<1>"):
        if name in summary["FETCH_SIZE"]:
            tr["assemble_bytes_per_launch"] = (2.0 * summary["FETCH_SIZE"][name]["avg_raw_kib"] + summary["WRITE_SIZE"].get(name, {}).get("avg_raw_kib", 0.0)) * 1024.0
    json.dump(tr, open(os.path.join(out, "traffic_%s.json" % wl), "w"), indent=1)
open(os.path.join(out, "summary_%s_%s.txt" % (tag, wl)), "w").write("\n".join(lines) + "\n")
json.dump(summary, open(os.path.join(out, "summary_%s_%s.json" % (tag, wl)), "w"), indent=1)
print("\n".join(lines))

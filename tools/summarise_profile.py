#!/usr/bin/env python3
"""Collapse a tools/profile_step.sh output directory into summary.json (per-launch means for the step kernel)."""
import collections
import csv
import glob
import json
import os
import sys

d = sys.argv[1]
out = {"counters_per_launch": {}, "kernel_stats": []}
# the step kernel's launches of the bench's main loop: the first (warm-up + timed) full-size launches in dispatch
# order -- the side measurements (carved pool, actor loop at 262,144 boards) come later or have another grid
for f in sorted(glob.glob(os.path.join(d, "pmc_*", "*", "*_counter_collection.csv"))):
    acc = collections.defaultdict(list)
    rows = [r for r in csv.DictReader(open(f)) if "step_kernel" in r["Kernel_Name"]]
    if not rows:
        continue
    full = max(int(r["Grid_Size"]) for r in rows)
    rows = sorted((r for r in rows if int(r["Grid_Size"]) == full), key=lambda r: int(r["Dispatch_Id"]))
    for row in rows:
        acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, v in acc.items():
        v = v[10:70] if len(v) >= 70 else v
        out["counters_per_launch"][k] = sum(v) / len(v)
for f in glob.glob(os.path.join(d, "kt", "*", "*_kernel_stats.csv")):
    for row in csv.DictReader(open(f)):
        if row["Name"].startswith(("void tpl::", "tpl::")):
            out["kernel_stats"].append({k: row[k] for k in ("Name", "Calls", "AverageNs", "MinNs", "MaxNs", "Percentage")})
# duration of the same launches, from the kernel trace (the stats file averages every size the bench launches)
for f in glob.glob(os.path.join(d, "kt", "*", "*_kernel_trace.csv")):
    rows = [r for r in csv.DictReader(open(f)) if "step_kernel" in r["Kernel_Name"]]
    if rows:
        full = max(int(r["Grid_Size_X"]) for r in rows)
        rows = sorted((r for r in rows if int(r["Grid_Size_X"]) == full), key=lambda r: int(r["Dispatch_Id"]))
        rows = rows[10:70] if len(rows) >= 70 else rows
        ns = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows]
        gaps = [int(b["Start_Timestamp"]) - int(a["End_Timestamp"]) for a, b in zip(rows, rows[1:])]
        out["step_kernel_main_loop"] = {"kernel": rows[0]["Kernel_Name"], "grid": full, "launches": len(ns),
                                        "average_ns": sum(ns) / len(ns), "min_ns": min(ns), "max_ns": max(ns),
                                        "median_gap_to_next_launch_ns": sorted(gaps)[len(gaps) // 2] if gaps else None}
c = out["counters_per_launch"]
if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
    out["hbm_bytes_per_launch_uncorrected"] = (c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024
    out["hbm_bytes_per_launch_fetch_doubled"] = (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024
json.dump(out, open(os.path.join(d, "summary.json"), "w"), indent=1)
print(json.dumps(out, indent=1))

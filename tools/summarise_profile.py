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
for f in sorted(glob.glob(os.path.join(d, "pmc_*", "*", "*_counter_collection.csv"))):
    acc = collections.defaultdict(list)
    for row in csv.DictReader(open(f)):
        if "step_kernel" in row["Kernel_Name"] or "rollout_kernel" in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, v in acc.items():
        v = v[10:] or v
        out["counters_per_launch"][k] = sum(v) / len(v)
for f in glob.glob(os.path.join(d, "kt", "*", "*_kernel_stats.csv")):
    for row in csv.DictReader(open(f)):
        if row["Name"].startswith(("void tpl::", "tpl::")):
            out["kernel_stats"].append({k: row[k] for k in ("Name", "Calls", "AverageNs", "MinNs", "MaxNs", "Percentage")})
c = out["counters_per_launch"]
if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
    out["hbm_bytes_per_launch_uncorrected"] = (c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024
    out["hbm_bytes_per_launch_fetch_doubled"] = (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024
json.dump(out, open(os.path.join(d, "summary.json"), "w"), indent=1)
print(json.dumps(out, indent=1))

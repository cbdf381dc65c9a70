#!/usr/bin/env python3
"""Collapse a rocprofv3 output directory (tools/profile_*.sh: `kt/` = --kernel-trace --stats, `pmc_*/` = one counter
pass each) into summary.json.

* `stamp`: the commit and source digest the profile was taken on (tools/profile_stamp.py).
* `kernels`: duration per (kernel, grid size) from the kernel trace -- rocprofv3's own kernel_stats.csv averages every
  launch of a template instance together, so a 2^23-board launch would drown the 2^20-board ones.
* `counters`: per (kernel, grid size), the mean of each counter per launch.
* `step_kernel_main_loop` (when the step kernel is in the trace): the launches of the bench's timed loop = the grid size
  with the most launches; `traffic` by the guide's formula and `traffic_decomposed` beside it.

Usage: tools/summarise_profile.py <dir> [kernel-name-substring ...]   (default: every tpl:: kernel)
"""
import collections
import csv
import glob
import json
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from profile_stamp import stamp  # noqa: E402

d = sys.argv[1]
want = sys.argv[2:] or ["tpl::"]


def short(name):
    name = name.replace("(anonymous namespace)::", "").split("(")[0]
    return name[5:] if name.startswith("void ") else name


def wanted(name):
    return "tpl::" in name and any(w in name for w in want)


out = {"stamp": stamp(), "kernels": [], "counters": {}}

trace = collections.defaultdict(list)
for f in glob.glob(os.path.join(d, "kt", "*", "*_kernel_trace.csv")):
    for r in csv.DictReader(open(f)):
        if wanted(r["Kernel_Name"]):
            trace[(short(r["Kernel_Name"]), int(r["Grid_Size_X"]), int(r["Workgroup_Size_X"]))].append(
                (int(r["Dispatch_Id"]), int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
for (name, grid, wg), rows in sorted(trace.items(), key=lambda kv: -sum(e - s for _, s, e in kv[1])):
    rows.sort()
    ns = [e - s for _, s, e in rows]
    steady = ns[2:] if len(ns) > 4 else ns                     # the first launches load code and touch cold pages
    out["kernels"].append({"name": name, "grid": grid, "workgroup": wg, "calls": len(ns), "total_ns": sum(ns),
                           "average_ns": sum(steady) / len(steady), "median_ns": statistics.median(steady),
                           "min_ns": min(ns), "max_ns": max(ns)})

counters = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(os.path.join(d, "pmc_*", "*", "*_counter_collection.csv"))):
    for r in csv.DictReader(open(f)):
        if wanted(r["Kernel_Name"]):
            counters[f"{short(r['Kernel_Name'])} grid={int(r['Grid_Size'])}"][r["Counter_Name"]].append(
                (int(r["Dispatch_Id"]), float(r["Counter_Value"])))
for key, cs in counters.items():
    for c, v in cs.items():
        v = [x for _, x in sorted(v)]
        v = v[2:] if len(v) > 4 else v
        out["counters"].setdefault(key, {})[c] = sum(v) / len(v)
    out["counters"][key]["launches"] = max(len(v) for v in cs.values())

# the step kernel's main loop: the (instance, grid) with the most launches
steps = [k for k in out["kernels"] if "step_kernel" in k["name"]]
if steps:
    main = max(steps, key=lambda k: k["calls"])
    rows = sorted(trace[(main["name"], main["grid"], main["workgroup"])])
    gaps = [b[1] - a[2] for a, b in zip(rows, rows[1:])]
    sec = {"kernel": main["name"], "grid": main["grid"], "launches": main["calls"], "average_ns": main["average_ns"],
           "median_ns": main["median_ns"], "min_ns": main["min_ns"], "max_ns": main["max_ns"],
           "median_gap_to_next_launch_ns": statistics.median(gaps) if gaps else None}
    c = out["counters"].get(f"{main['name']} grid={main['grid']}", {})
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        boards = main["grid"] * (2 if ", 2, " in main["name"] else 1)           # boards per lane is a template argument
        sec["boards_per_launch"] = boards
        # guide formula: FETCH_SIZE tallies every read request at 64 B; a 128-B request is two of them -> doubled
        sec["traffic"] = (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024
        sec["traffic_uncorrected"] = (c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024
        req = c.get("TCC_EA0_RDREQ_sum")
        if req:
            # the coalesced streams (32 B state + 1 B action per board) arrive as 128-B requests; every other read
            # request is a 64-B gather (a finished board's pool record, a window refill)
            stream = min(req, (32 + 1) * boards / 128.0)
            sec["read_requests"] = req
            sec["traffic_decomposed"] = stream * 128 + (req - stream) * 64 + c["WRITE_SIZE"] * 1024
            sec["decomposition"] = {"stream_requests_128B": stream, "gather_requests_64B": req - stream,
                                    "write_bytes": c["WRITE_SIZE"] * 1024}
        for k in ("traffic", "traffic_decomposed"):
            if k in sec:
                sec[k + "_GBs"] = sec[k] / main["average_ns"]
                sec["frac_" + k] = sec[k] / main["average_ns"] / 8000.0
        sec["algorithmic_bytes"] = 96 * boards
        sec["frac"] = 96 * boards / main["average_ns"] / 8000.0
    out["step_kernel_main_loop"] = sec

c = {k: v for k, v in out["counters"].items() if "SQ_THREAD_CYCLES_VALU" in v and v.get("SQ_ACTIVE_INST_VALU")}
for key, v in c.items():
    v["lanes_active_per_valu_instruction"] = v["SQ_THREAD_CYCLES_VALU"] / v["SQ_ACTIVE_INST_VALU"]
    if v.get("SQ_WAVE_CYCLES"):
        v["valu_active_share_of_wave_cycles"] = v["SQ_ACTIVE_INST_VALU"] / v["SQ_WAVE_CYCLES"]

json.dump(out, open(os.path.join(d, "summary.json"), "w"), indent=1)
print(json.dumps({k: out[k] for k in out if k != "counters"}, indent=1)[:6000])

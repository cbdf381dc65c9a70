#!/bin/bash
# Three independent readings of what the chip does under the bf16 policy kernel, on one box in one call:
#   1. the driver's own telemetry (sysfs: shader clock, board power against its cap) sampled during seconds of launches
#   2. the in-kernel clock (diagnostic build: s_memtime over s_memrealtime around each wave's tile loop)
#   3. GRBM_GUI_ACTIVE / 8 XCDs / kernel duration on a dispatch of 2^23 boards (~1 ms), counters and duration in passes of their own
# Usage: tools/policy_evidence.sh <outdir under gpurun_out>
OUT=gpurun_out/${1:-policy_evidence}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 120 python3 tools/policy_power_trace.py --seconds 3 > $OUT/power_trace.jsonl 2> $OUT/power_trace.err
TPL_DIAG_CLOCK=1 timeout -k 10 300 python3 tools/policy_clock.py > $OUT/in_kernel_clock.log 2>&1
B="python3 tools/policy_probe.py 8388608"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- $B > $OUT/kt.log 2>&1 || echo "kt failed"
timeout -k 10 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc -- $B > $OUT/pmc.log 2>&1 || echo "pmc failed"
python3 - $OUT <<'PY'
import csv, glob, json, os, sys
sys.path.insert(0, "tools")
from profile_stamp import stamp
d = sys.argv[1]
out = {"stamp": stamp()}
for f in glob.glob(os.path.join(d, "kt", "*", "*_kernel_stats.csv")):
    for row in csv.DictReader(open(f)):
        if "policy_kernel" in row["Name"]:
            out["kernel_stats_8M_boards"] = {k: row[k] for k in ("Name", "Calls", "AverageNs", "MinNs", "MaxNs")}
acc = {}
for f in glob.glob(os.path.join(d, "pmc", "*", "*_counter_collection.csv")):
    for row in csv.DictReader(open(f)):
        if "policy_kernel" in row["Kernel_Name"]:
            acc.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
out["counters_per_launch_8M_boards"] = {k: sum(v[2:]) / max(1, len(v[2:])) for k, v in acc.items()}
ks, c = out.get("kernel_stats_8M_boards"), out["counters_per_launch_8M_boards"]
if ks and "GRBM_GUI_ACTIVE" in c:
    ns = float(ks["AverageNs"])
    out["effective_clock_ghz_grbm_over_8_over_duration"] = c["GRBM_GUI_ACTIVE"] / 8 / ns
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
        out["mfma_busy_fraction_of_active_cycles"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / (c["GRBM_GUI_ACTIVE"] / 8)
json.dump(out, open(os.path.join(d, "summary.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
PY
head -c 400 $OUT/power_trace.jsonl; echo; cat $OUT/in_kernel_clock.log | tail -2
# the raw per-dispatch tables are large (gpurun brings back 64 MiB at most): the summary holds what is kept
find $OUT -name "*_counter_collection.csv" -delete; find $OUT -name "*_kernel_trace.csv" -delete; find $OUT -name "*.db" -delete

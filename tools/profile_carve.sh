#!/bin/bash
# rocprofv3 on the device carving generator: kernel trace + stats, then counter passes of their own (duration, how busy
# the vector ALU is, and how many of a wave's 64 lanes its instructions run on).  Usage: tools/profile_carve.sh <tag> [probe args]
set -u
TAG=${1:-carve}; shift || true
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B="python3 tools/carve_probe.py $*"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- $B > $OUT/kt.log 2>&1 || echo "kt failed"
for P in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU" \
         "SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE"; do
  tag=$(echo $P | cut -d" " -f1)
  timeout -k 10 200 rocprofv3 --pmc $P --output-format csv -d $OUT/pmc_$tag -- $B > $OUT/pmc_$tag.log 2>&1 || echo "pmc $tag failed"
done
python3 - $OUT <<'PY'
import collections, csv, glob, json, os, sys
d = sys.argv[1]
out = {"kernel_stats": [], "counters_per_launch": {}}
for f in glob.glob(os.path.join(d, "kt", "*", "*_kernel_stats.csv")):
    for row in csv.DictReader(open(f)):
        if "carve_kernel" in row["Name"]:
            out["kernel_stats"].append({k: row[k] for k in ("Name", "Calls", "AverageNs", "MinNs", "MaxNs", "Percentage")})
for f in sorted(glob.glob(os.path.join(d, "pmc_*", "*", "*_counter_collection.csv"))):
    acc = collections.defaultdict(list)
    rows = [r for r in csv.DictReader(open(f)) if "carve_kernel" in r["Kernel_Name"]]
    if not rows:
        continue
    full = max(int(r["Grid_Size"]) for r in rows)                  # the probe's full-size launches, not its 4096 warm-up
    for r in rows:
        if int(r["Grid_Size"]) == full:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        out["counters_per_launch"][k] = sum(v) / len(v)
c = out["counters_per_launch"]
if "SQ_THREAD_CYCLES_VALU" in c and c.get("SQ_ACTIVE_INST_VALU"):
    out["thread_cycles_per_active_valu_cycle"] = c["SQ_THREAD_CYCLES_VALU"] / c["SQ_ACTIVE_INST_VALU"]
if "SQ_ACTIVE_INST_VALU" in c and c.get("SQ_WAVE_CYCLES"):
    out["valu_active_share_of_wave_cycles"] = c["SQ_ACTIVE_INST_VALU"] / c["SQ_WAVE_CYCLES"]
json.dump(out, open(os.path.join(d, "summary.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
PY

#!/bin/bash
# PMC profile of the fused policy kernel (separate passes).  Usage: tools/profile_policy.sh <tag>
set -u
TAG=${1:-policy}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B="python3 tools/actor_probe.py"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- $B > $OUT/kt.log 2>&1 || echo "kt failed"
for P in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
         "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"; do
  tag=$(echo $P | cut -d" " -f1)
  timeout -k 10 200 rocprofv3 --pmc $P --output-format csv -d $OUT/pmc_$tag -- $B > $OUT/pmc_$tag.log 2>&1 || echo "pmc $tag failed"
done
python3 - "$OUT" <<'PY'
import collections, csv, glob, json, os, sys
sys.path.insert(0, "tools")
from profile_stamp import stamp
d = sys.argv[1]
out = {"stamp": stamp()}
for f in sorted(glob.glob(os.path.join(d, "pmc_*", "*", "*_counter_collection.csv"))):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(f)):
        if "policy_kernel" in row["Kernel_Name"]:
            acc[row["Kernel_Name"][:40]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, cs in acc.items():
        for c, v in cs.items():
            v = v[5:] or v
            out.setdefault(k, {})[c] = sum(v) / len(v)
for f in glob.glob(os.path.join(d, "kt", "*", "*_kernel_stats.csv")):
    for row in csv.DictReader(open(f)):
        if "policy_kernel" in row["Name"] or "step_kernel" in row["Name"]:
            out.setdefault("kernel_stats", []).append({k: row[k] for k in ("Name", "Calls", "AverageNs", "MinNs", "MaxNs")})
json.dump(out, open(os.path.join(d, "summary.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
PY
# the raw per-dispatch tables are large (gpurun brings back 64 MiB at most): the summary holds what is kept
find $OUT -name "*_counter_collection.csv" -delete; find $OUT -name "*_kernel_trace.csv" -delete; find $OUT -name "*.db" -delete

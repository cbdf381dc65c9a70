#!/bin/bash
# rocprofv3 kernel stats + PMC for the fused rollout kernel and the policy kernels (separate passes).
set -u
OUT=gpurun_out/prof_${1:-r02_final}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B="python3 bench.py --steps 200 --warmup 10 --no-cpu-baseline --chunk 50 --carved-pool 0 --no-config1 --sustained 0"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- $B > $OUT/kt.log 2>&1 || echo "kt failed"
for P in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
         "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
         "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $P | cut -d" " -f1)
  timeout -k 10 300 rocprofv3 --pmc $P --output-format csv -d $OUT/pmc_$tag -- $B > $OUT/pmc_$tag.log 2>&1 || echo "pmc $tag failed"
done
python3 - "$OUT" <<'PY'
import collections, csv, glob, json, os, sys
d = sys.argv[1]
out = {"counters_per_launch": {}, "kernel_stats": []}
for f in sorted(glob.glob(os.path.join(d, "pmc_*", "*", "*_counter_collection.csv"))):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"]
        if "tpl::" in name and any(k in name for k in ("step_kernel", "rollout_kernel", "policy_kernel", "policy_f32_kernel")):
            acc[name.split("(")[0][-60:]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, cs in acc.items():
        for c, v in cs.items():
            v = v[2:] or v
            out["counters_per_launch"].setdefault(k, {})[c] = sum(v) / len(v)
for f in glob.glob(os.path.join(d, "kt", "*", "*_kernel_stats.csv")):
    for row in csv.DictReader(open(f)):
        if "tpl::" in row["Name"]:
            out["kernel_stats"].append({k: row[k] for k in ("Name", "Calls", "AverageNs", "MinNs", "MaxNs", "Percentage")})
json.dump(out, open(os.path.join(d, "summary.json"), "w"), indent=1)
print(json.dumps(out["kernel_stats"], indent=1))
PY

#!/bin/bash
# rocprofv3 summary of the policy path's kernels on the loaded library (tools/policy_probe.py at 262,144 boards): durations from a
# kernel trace, counters in separate --pmc passes (never combined with a trace domain), and per kernel the figures derived from
# them: the clock held (GRBM_GUI_ACTIVE / 8 XCDs / duration), the MFMA-busy share of the active cycles, LDS bank-conflict cycles
# per LDS-active cycle.  Usage: tools/profile_policy.sh <tag>  ->  gpurun_out/prof_<tag>/summary.json
set -u
TAG=${1:-policy}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B="python3 tools/policy_probe.py 262144"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- $B > $OUT/kt.log 2>&1 || echo "kt failed"
for P in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
         "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" \
         "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" \
         "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM"; do
  tag=$(echo $P | cut -d" " -f1)
  timeout -k 10 240 rocprofv3 --pmc $P --output-format csv -d $OUT/pmc_$tag -- $B > $OUT/pmc_$tag.log 2>&1 || echo "pmc $tag failed"
done
python3 tools/summarise_profile.py $OUT policy_kernel policy_f32_kernel policy_split_kernel actor_rollout > /dev/null
python3 - "$OUT" <<'PY'
import json, os, sys
d = sys.argv[1]
s = json.load(open(os.path.join(d, "summary.json")))
FLOPS, BOARDS = 2.0 * (217 * 128 + 3 * 128 * 128 + 128 * 14), 262144
derived = {}
for k in s["kernels"]:
    c = s["counters"].get(f"{k['name']} grid={k['grid']}", {})
    ns = k["average_ns"]
    row = {"average_ns": ns, "calls": k["calls"]}
    if "GRBM_GUI_ACTIVE" in c:
        active = c["GRBM_GUI_ACTIVE"] / 8.0                                  # the counter sums the 8 XCDs
        row["clock_GHz_held"] = active / ns
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
            row["mfma_busy_share_of_active_cycles"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / active     # 1,024 SIMDs
    for name in ("SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_INSTS_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_VALU_MFMA_BUSY_CYCLES"):
        if name in c:
            row[name] = c[name]
    if c.get("SQ_LDS_IDX_ACTIVE"):
        row["lds_bank_conflict_cycles_per_lds_active_cycle"] = c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"]
    if "actor_rollout" not in k["name"]:
        row["model_TFLOPs"] = FLOPS * BOARDS / ns / 1e3
    derived[f"{k['name']} grid={k['grid']}"] = row
s["derived"] = derived
s["workload"] = "tools/policy_probe.py 262144: BASELINE configs[4]'s policy, Model(217, 14), random-init weights, synthetic boards"
json.dump(s, open(os.path.join(d, "summary.json"), "w"), indent=1)
print(json.dumps(derived, indent=1))
PY
grep -h "boards:" $OUT/kt.log
# the raw per-dispatch tables are large (gpurun brings back 64 MiB at most): the summary holds what is kept
find $OUT -name "*_counter_collection.csv" -delete; find $OUT -name "*_kernel_trace.csv" -delete; find $OUT -name "*.db" -delete

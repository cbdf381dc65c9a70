#!/bin/bash
# Vector / scalar / LDS instructions and wave cycles of one fused-rollout launch (one rocprofv3 counter pass):
#   tools/rollout_valu_count.sh <tag> [rollout_probe args]
TAG=${1:-x}; shift || true
OUT=gpurun_out/rvalu_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_THREAD_CYCLES_VALU SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $OUT -- python3 tools/rollout_probe.py --launches 6 "$@" > $OUT.log 2>&1 || echo "pmc failed"
python3 - $OUT <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "rollout_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print({k: f"{sum(v[2:]) / len(v[2:]):.4g}" for k, v in sorted(acc.items())})
PY
rm -rf $OUT

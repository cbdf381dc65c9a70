#!/bin/bash
# Vector and scalar instructions a launch of the device carving generator executes (one rocprofv3 counter pass):
#   tools/carve_valu_count.sh <tag> [carve_probe args]     -> gpurun_out/valu_<tag>.txt
TAG=${1:-x}; shift || true
OUT=gpurun_out/valu_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_THREAD_CYCLES_VALU --output-format csv -d $OUT -- python3 tools/carve_probe.py --launches 3 "$@" > $OUT.log 2>&1 || echo "pmc failed"
python3 - $OUT <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "carve_kernel" in r["Kernel_Name"] and int(r["Grid_Size"]) > 100000:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print({k: f"{sum(v) / len(v):.4g}" for k, v in sorted(acc.items())})
PY
rm -rf $OUT

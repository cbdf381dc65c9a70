#!/bin/bash
# Instruction counters of the rollout kernel, per tree: tools/pmc_rollout.sh <outdir> <tree> [<tree> ...]
OUT=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ROOT=$PWD
for tree in "$@"; do
  name=$(echo $tree | tr '/.' '__')
  cd $ROOT/$tree
  python3 $ROOT/tools/rollout_probe.py 2>/dev/null | tail -1
  for P in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"; do
    tag=$(echo $P | cut -d" " -f1)
    timeout -k 10 200 rocprofv3 --pmc $P --output-format csv -d $ROOT/$OUT/${name}_$tag -- python3 $ROOT/tools/rollout_probe.py --launches 4 > $ROOT/$OUT/${name}_$tag.log 2>&1 || echo "pmc $tag failed"
  done
  cd $ROOT
  python3 - $OUT $name <<'PY'
import collections, csv, glob, os, sys
d, name = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for f in glob.glob(os.path.join(d, name + "_*", "*", "*_counter_collection.csv")):
    for row in csv.DictReader(open(f)):
        if "rollout_kernel" in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
print(name, {k: sum(v[2:]) / max(1, len(v[2:])) for k, v in sorted(acc.items())})
PY
done

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
         "SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE" \
         "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  tag=$(echo $P | cut -d" " -f1)
  timeout -k 10 200 rocprofv3 --pmc $P --output-format csv -d $OUT/pmc_$tag -- $B > $OUT/pmc_$tag.log 2>&1 || echo "pmc $tag failed"
done
python3 tools/summarise_profile.py $OUT carve_kernel
# the raw per-dispatch tables are large (gpurun brings back 64 MiB at most): the summary holds what is kept
find $OUT -name "*_counter_collection.csv" -delete; find $OUT -name "*_kernel_trace.csv" -delete; find $OUT -name "*.db" -delete

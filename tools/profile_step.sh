#!/bin/bash
# Profile the bench workload with rocprofv3: kernel trace + stats, then PMC counters in separate passes
# (gpurun refuses --pmc combined with trace domains).  Usage: tools/profile_step.sh <tag> [bench args...]
set -u
TAG=${1:-run}; shift || true
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B="python3 bench.py --steps 60 --warmup 10 --no-cpu-baseline $*"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- $B > $OUT/kt.log 2>&1 || echo "kt failed"
for P in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
         "FETCH_SIZE" "WRITE_SIZE" \
         "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" \
         "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM GRBM_GUI_ACTIVE" \
         "TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_64B_sum TCC_REQ_sum TCC_READ_sum" \
         "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCC_WRITE_sum TCC_ATOMIC_sum"; do
  tag=$(echo $P | cut -d" " -f1)
  timeout -k 10 200 rocprofv3 --pmc $P --output-format csv -d $OUT/pmc_$tag -- $B > $OUT/pmc_$tag.log 2>&1 || echo "pmc $tag failed"
done
python3 tools/summarise_profile.py $OUT
# the raw per-dispatch tables are large (gpurun brings back 64 MiB at most): the summary holds what is kept
find $OUT -name "*_counter_collection.csv" -delete; find $OUT -name "*_kernel_trace.csv" -delete; find $OUT -name "*.db" -delete

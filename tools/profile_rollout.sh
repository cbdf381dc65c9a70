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
python3 tools/summarise_profile.py $OUT step_kernel rollout_kernel policy_kernel policy_f32_kernel actor_rollout
# the raw per-dispatch tables are large (gpurun brings back 64 MiB at most): the summary holds what is kept
find $OUT -name "*_counter_collection.csv" -delete; find $OUT -name "*_kernel_trace.csv" -delete; find $OUT -name "*.db" -delete

#!/bin/bash
# The counters behind bench.py's `valu-issue` rooflines (profiles/valu_issue.json): for every kernel that HBM does not bound, in
# exactly the form the bench runs it, one kernel-trace pass (durations) and one counter pass (vector instructions, the lanes
# they ran on, the shader clock):   tools/profile_valu.sh <tag>   ->   gpurun_out/prof_<tag>/<form>/summary.json
# then, in the build container:  python tools/update_valu_issue.py gpurun_out/prof_<tag> profiles/<dir>
set -u
TAG=${1:-valu}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run() {   # run <form> <kernel substring> <probe command ...>
  local form=$1 kern=$2; shift 2
  local D=$OUT/$form
  mkdir -p $D
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $D/kt -- "$@" > $D/kt.log 2>&1 || echo "$form: kt failed"
  timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE \
      --output-format csv -d $D/pmc_valu -- "$@" > $D/pmc.log 2>&1 || echo "$form: pmc failed"
  python3 tools/summarise_profile.py $D $kern > /dev/null
  find $D -name "*_counter_collection.csv" -delete; find $D -name "*_kernel_trace.csv" -delete; find $D -name "*.db" -delete
  tail -1 $D/kt.log
}
run rollout_f32_u8_50      rollout_kernel python3 tools/rollout_probe.py --steps-per-launch 50 --launches 8 --outputs 1
run rollout_compact_50     rollout_kernel python3 tools/rollout_probe.py --steps-per-launch 50 --launches 8 --outputs 2
run rollout_random_100     rollout_kernel python3 tools/rollout_probe.py --steps-per-launch 100 --launches 6 --random
run rollout_shard_131072   rollout_kernel python3 tools/rollout_probe.py --boards 131072 --steps-per-launch 50 --launches 12 --outputs 1
run carve_1048576          carve_kernel   python3 tools/carve_probe.py --count 1048576 --launches 4
run carve_262144           carve_kernel   python3 tools/carve_probe.py --count 262144 --launches 4

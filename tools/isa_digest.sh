#!/bin/bash
# One digest over the disassembly of every kernel of a built library (addresses stripped): two trees whose device code is the
# same print the same line.  tools/isa_digest.sh [lib]   (no GPU needed; the round-6 freeze was checked with it against a
# `git worktree` of round 5's last commit: profiles/r06_final/kernel_isa_digest.log)
LIB=${1:-lib/libtetris_piclim.so}
cd "$(dirname "$0")/.."
for k in step_kernel rollout_kernel carve_kernel reset_kernel export_kernel observe policy_kernel policy_f32_kernel policy_split_kernel actor_rollout forward; do
  tools/dump_isa.sh $k $LIB | sed 's/^ *[0-9a-f]*://'
done | sha256sum | cut -d" " -f1

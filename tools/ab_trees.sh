#!/bin/bash
# Same-box A/B of a probe across source trees: tools/ab_trees.sh "<probe command>" tree [tree ...]   (alternating, three rounds)
# e.g. tools/ab_trees.sh "python tools/policy_probe.py" _ab/r3 _ab/old .      (_ab/* = git worktrees of earlier commits, built here)
CMD=$1; shift
ROOT=$PWD
for round in 1 2 3; do
  for tree in "$@"; do
    echo -n "$tree round $round: "
    (cd $ROOT/$tree && $CMD 2>&1 | grep -v amdgpu.ids | tail -1)
  done
done

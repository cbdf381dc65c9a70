#!/bin/bash
# A/B on ONE box: the bench's sustained step-kernel figure of the tree in _ab/old against the working tree, alternating.
ARGS="--no-cpu-baseline --actor-boards 0 --carved-pool 0 --no-config1 --steps 500 --warmup 50"
for round in 1 2 3; do
  for tree in _ab/old .; do
    [ -f $tree/bench.py ] || continue
    (cd $tree && python bench.py $ARGS 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); s=d['roofline']['sustained']; f=d['fused_rollout']
        print('$tree round $round: sustained median %.3f us  mean %.3f us  timed %.3f us  fused %.2f us (%.1f G)' % (s['kernel_ms_median_of_50s']*1e3, s['kernel_ms_mean']*1e3, d['ms_per_step']*1e3, f['ms_per_step']*1e3, f['value']/1e9))
")
  done
done

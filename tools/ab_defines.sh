#!/bin/bash
# A/B on ONE box of source variants of the working tree: tools/ab_defines.sh "" TPL_X_FOO TPL_X_BAR ...  (each name is a
# -D define; _lib.py builds lib/libtetris_piclim_<NAME>.so for it).  "old" = the tree in _ab/old.  Alternating rounds.
ARGS=${AB_ARGS:-"--no-cpu-baseline --actor-boards 0 --carved-pool 0 --no-config1 --steps 500 --warmup 50"}
for round in 1 2 3; do
  for v in "$@"; do
    tree=.; def=$v
    if [ "$v" = "old" ]; then tree=_ab/old; def=""; fi
    (cd $tree && TPL_EXTRA_DEFINE=$def python bench.py $ARGS 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); s=d['roofline']['sustained']; f=d['fused_rollout']
        print('%-16s round $round: sustained median %.3f us  mean %.3f us  timed %.3f us  fused %.2f us (%.1f G)' % ('${v:-default}', s['kernel_ms_median_of_50s']*1e3, s['kernel_ms_mean']*1e3, d['ms_per_step']*1e3, f['ms_per_step']*1e3, f['value']/1e9))
")
  done
done

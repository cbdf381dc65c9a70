#!/bin/bash
# A/B on ONE box of source variants of the working tree: tools/ab_defines.sh "" TPL_X_FOO TPL_X_BAR ...  (each name is a
# -D define; _lib.py builds lib/libtetris_piclim_<NAME>.so for it).  "old" = the tree in _ab/old (a `git worktree` of an earlier
# commit; a tree from before round 6 prints its whole record on stdout and is read from there).  Alternating rounds of the bench's
# sustained step figure and fused rollout.  The rule since round 5: no change to tpl_device.h, step_kernel,
# rollout_kernel or carve_kernel without >= 5 % here and a green `pytest -m gpu`.
ARGS=${AB_ARGS:-"--no-cpu-baseline --actor-boards 0 --carved-pool 0 --no-config1 --no-out-of-cache --shard-ranks 0 --steps 500 --warmup 50"}
for round in 1 2 3; do
  for v in "$@"; do
    tree=.; def=$v
    if [ "$v" = "old" ]; then tree=_ab/old; def=""; fi
    (cd $tree && TPL_EXTRA_DEFINE=$def python bench.py $ARGS --detail /tmp/ab_detail.json 2>/dev/null | python -c "
import json,os,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l)
        if 'sustained' not in d['roofline'] and os.path.exists('/tmp/ab_detail.json'): d=json.load(open('/tmp/ab_detail.json'))
        s=d['roofline']['sustained']; f=d['fused_rollout']
        print('%-16s round $round: sustained median %.3f us  mean %.3f us  timed %.3f us  fused %.2f us (%.1f G)' % ('${v:-default}', s['kernel_ms_median_of_50s']*1e3, s['kernel_ms_mean']*1e3, d['ms_per_step']*1e3, f['ms_per_step']*1e3, f['value']/1e9))
"; rm -f /tmp/ab_detail.json)
  done
done

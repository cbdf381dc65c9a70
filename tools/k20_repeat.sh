for r in 1 2 3; do for args in "--no-cpu-baseline --actor-boards 0 --carved-pool 0 --no-config1" "--no-cpu-baseline"; do python bench.py --steps 20 --warmup 5 $args 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$args'[:40].ljust(40), 'timed %.3f us/step, launch after the synchronize %.1f us, sustained %.3f' % (d['ms_per_step']*1e3, (d['timing']['launch_after_synchronize_ms'] or 0)*1e3, d['roofline']['sustained']['kernel_ms_median_of_50s']*1e3))
"; done; done

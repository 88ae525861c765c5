#!/bin/bash
# optimistic schedule with row chunks in place: is the 393 216-row cap of a launch still worth a pool select?  (ablation library)
export LDOT_LIBRARY=$PWD/lightningdot_amd/libldot_ablation.so
run() {
  env "$@" timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print('$*: ms/step %.3f  kernel_ms/step %.3f launches %.0f  frac %.3f recall@1 %.3f redone %d' % (d['ms_per_step'], r['kernel_ms_per_step'], r['launches_per_step'], r['frac'], d['recall@1'], d['overflowed_queries']), flush=True)
"
}
for rep in 1 2; do
run X=default
run LDOT_DEBUG_OPT_MAXROWS=1000000
run LDOT_DEBUG_OPT_MAXROWS=1000000 LDOT_DEBUG_OPT_GROWTHX=10
run LDOT_DEBUG_OPT_MAXROWS=589824
run LDOT_DEBUG_OPT_GROWTHX=5
done

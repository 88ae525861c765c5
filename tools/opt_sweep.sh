#!/bin/bash
# sweep of the optimistic-threshold schedule (ablation library): tools/opt_sweep.sh
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
run LDOT_DEBUG_OPT_EPS=1e-5
run LDOT_DEBUG_OPT_EPS=1e-9
run LDOT_DEBUG_OPT_GROWTHX=5
run LDOT_DEBUG_OPT_GROWTHX=11
run LDOT_DEBUG_OPT_GROWTHX=15
run LDOT_DEBUG_OPT_MAXROWS=294912
run LDOT_DEBUG_OPT_MAXROWS=491520
run LDOT_DEBUG_OPT_MAXROWS=1000000
done

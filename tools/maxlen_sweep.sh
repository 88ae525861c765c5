#!/bin/bash
# Launches capped to row ranges that fit the Infinity Cache (ablation library, guaranteed schedule, tau = +inf so that admissions do not move):
# does the row stream of the 2nd..5th query group of a launch come cheaper when the launch's rows are still cached?  tools/maxlen_sweep.sh
export LDOT_LIBRARY=$PWD/lightningdot_amd/libldot_ablation.so
for rep in 1 2; do
for m in 0 196608 98304 49152; do
  LDOT_DEBUG_NOOPT=1 LDOT_DEBUG_VARIANT=16 LDOT_DEBUG_MAXLEN=$m timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print('rep $rep maxlen $m: kernel_ms/step %.3f launches %.0f  ms/step %.3f' % (r['kernel_ms_per_step'], r['launches_per_step'], d['ms_per_step']), flush=True)
"
done
done

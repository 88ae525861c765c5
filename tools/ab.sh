#!/bin/bash
# usage: tools/ab.sh "<variants>" [rounds] [steps] ["extra bench args"]  -> kernel ms per step of every LDOT_DEBUG_VARIANT, interleaved over `rounds` rounds
# (ablation library: python -m lightningdot_amd.build --ablation -> lightningdot_amd/libldot_ablation.so)
export LDOT_LIBRARY=$PWD/lightningdot_amd/libldot_ablation.so
R=${2:-2}; S=${3:-10}; X=${4:-}
for r in $(seq 1 $R); do
for v in $1; do
  LDOT_DEBUG_VARIANT=$v timeout 300 python bench.py --steps $S --warmup 2 --no-cpu-baseline $X 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print('round $r variant $v $X: ms/step %.3f  kernel_ms/step %.3f launches %.0f  TF %.0f  recall@1 %.3f overflow %d' % (d['ms_per_step'], r['kernel_ms_per_step'], r['launches_per_step'], r['achieved'], d['recall@1'], d['overflowed_queries']), flush=True)
"
done
done

#!/bin/bash
# usage: tools/benchline.sh <label> [bench args]  -> one compact line
L=$1; shift
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print('$L: ms/step %.2f  q/s %.0f  kernel_ms %.2f  launches %.0f  TF %.0f  R@1 %.3f ovf %d' % (d['ms_per_step'], d['value'], r['kernel_ms_per_step'], r['launches_per_step'], r['achieved'], d['recall@1'], d['overflowed_queries']))
"

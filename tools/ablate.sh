#!/bin/bash
# usage: tools_ablate.sh "0 1 2 ..."   -> prints kernel ms per step for each LDOT_DEBUG_VARIANT
for v in $1; do
  LDOT_DEBUG_VARIANT=$v timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print('variant $v: ms/step %.2f  kernel_ms/step %.2f  TF %.0f  recall@1 %.3f overflow %d' % (d['ms_per_step'], r['kernel_ms_per_step'], r['achieved'], d['recall@1'], d['overflowed_queries']))
"
done

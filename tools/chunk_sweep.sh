#!/bin/bash
# chunk length of the fused scan's row chunks (ablation library, LDOT_DEBUG_CHUNK_TILES; a multiple of 32): tools/chunk_sweep.sh "<tiles ...>" [reps]
export LDOT_LIBRARY=$PWD/lightningdot_amd/libldot_ablation.so
for rep in $(seq 1 ${2:-2}); do
for c in $1; do
  LDOT_DEBUG_CHUNK_TILES=$c timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print('rep $rep chunk tiles $c: ms/step %.3f kernel_ms/step %.3f frac %.3f recall@1 %.3f overflow %d' % (d['ms_per_step'], r['kernel_ms_per_step'], r['frac'], d['recall@1'], d['overflowed_queries']), flush=True)
"
done
done

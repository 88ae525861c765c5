#!/bin/bash
# usage: tools/ab2.sh "<lib1> <lib2>" "<variants>" [rounds] [steps]: kernel ms per step of LDOT_DEBUG_VARIANT variants of SEVERAL ablation libraries,
# on the guaranteed schedule (--no-optimistic: the variants produce no candidates), interleaved inside one gpurun call
R=${3:-2}; S=${4:-10}
for r in $(seq 1 $R); do
for lib in $1; do
for v in $2; do
  LDOT_LIBRARY=$PWD/$lib LDOT_DEBUG_VARIANT=$v timeout 300 python bench.py --steps $S --warmup 2 --no-cpu-baseline --no-secondary --no-optimistic 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print('round $r $lib variant $v: ms/step %.3f  kernel_ms/step %.3f launches %.0f  TF %.0f' % (d['ms_per_step'], r['kernel_ms_per_step'], r['launches_per_step'], r['achieved']), flush=True)
"
done
done
done

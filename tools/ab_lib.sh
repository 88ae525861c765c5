#!/bin/bash
# usage: tools/ab_lib.sh "<lib1> <lib2> ..." [rounds] [steps]  -> bench.py of the headline workload with each library (LDOT_LIBRARY), interleaved
R=${2:-3}; S=${3:-20}
for r in $(seq 1 $R); do
for lib in $1; do
  LDOT_LIBRARY=$PWD/$lib timeout 300 python bench.py --steps $S --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print('round $r $lib: ms/step %.3f  kernel_ms/step %.3f  TF %.0f frac %.3f  q/s %.0f recall@1 %.3f overflow %d' % (d['ms_per_step'], r['kernel_ms_per_step'], r['achieved'], r['frac'], d['value'], d['recall@1'], d['overflowed_queries']), flush=True)
"
done
done

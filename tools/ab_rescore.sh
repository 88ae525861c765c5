#!/bin/bash
# usage: tools/ab_rescore.sh "<lib> ..." [rounds]: headline pass and COCO-shape evaluation with each library (LDOT_LIBRARY), interleaved
R=${2:-2}
for r in $(seq 1 $R); do
for lib in $1; do
  a=$(LDOT_LIBRARY=$PWD/lightningdot_amd/$lib timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'): d = json.loads(l); print('%.3f' % d['ms_per_step'])")
  b=$(LDOT_LIBRARY=$PWD/lightningdot_amd/$lib timeout 300 python bench.py --workload coco --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'): d = json.loads(l); print('%.3f' % d['ms_per_step'])")
  echo "round $r $lib: headline $a ms  coco $b ms"
done; done

#!/bin/bash
# A/B of the round-5 head (git archive 74cb2bf built in ab_r5/) against this tree on ONE box: headline bench lines interleaved, then the serving latencies
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06; mkdir -p $OUT
line() { python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print('$1: ms/step %.3f  kernel_ms/step %.3f  frac %.3f  q/s %.0f recall@1 %.3f' % (d['ms_per_step'], r['kernel_ms_per_step'], r['frac'], d['value'], d['recall@1']), flush=True)
"; }
( for r in 1 2 3; do
  (cd ab_r5 && timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | line "round $r r5  ")
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | line "round $r head"
done
echo "--- serving r5"; (cd ab_r5 && timeout 300 python tools/serving_latency.py 2>&1 | grep -v amdgpu.ids | tail -12)
echo "--- serving head"; timeout 300 python tools/serving_latency.py 2>&1 | grep -v amdgpu.ids | tail -12 ) > $OUT/ab_r5_vs_head.txt 2>&1
cat $OUT/ab_r5_vs_head.txt

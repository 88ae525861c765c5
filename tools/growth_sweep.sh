for rep in 1 2; do
for g in 100 150 200 250 300 400; do
  timeout 200 python bench.py --steps 15 --warmup 3 --no-cpu-baseline --growth $g 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print('rep $rep growth $g: ms/step %.3f kernel %.3f launches %.0f cand %s overflow %d' % (d['ms_per_step'], r['kernel_ms_per_step'], r['launches_per_step'], d.get('fused_candidates'), d['overflowed_queries']), flush=True)
"
done
done
for w in 2048 8192; do
  timeout 200 python bench.py --steps 15 --warmup 3 --no-cpu-baseline --warm $w 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print('warm $w: ms/step %.3f kernel %.3f launches %.0f overflow %d' % (d['ms_per_step'], r['kernel_ms_per_step'], r['launches_per_step'], d['overflowed_queries']), flush=True)
"
done

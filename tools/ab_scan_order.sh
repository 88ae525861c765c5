for r in 1 2 3; do
for cfg in "libldot_prev.so 0" "libldot.so 0" "libldot.so 2"; do
  set -- $cfg
  LDOT_LIBRARY=$PWD/lightningdot_amd/$1 timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary --scan-order $2 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print('round $r $1 scan-order $2: ms/step %.3f  kernel_ms/step %.3f launches %.0f  recall %.3f overflow %d' % (d['ms_per_step'], r['kernel_ms_per_step'], r['launches_per_step'], d['recall@1'], d['overflowed_queries']), flush=True)
"
done; done

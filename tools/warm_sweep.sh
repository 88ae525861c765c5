#!/bin/bash
# headline ms/step against the dense warm-up length (LDOT_OPT_WARM_ROWS), interleaved rounds in one call
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05x
for r in 1 2 3; do for w in 4096 5120 6144 8192; do
python bench.py --steps 15 --warmup 3 --no-cpu-baseline --no-secondary --warm $w 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print('round $r warm $w: ms/step %.3f kernel_ms %.3f launches %.0f' % (d['ms_per_step'], r['kernel_ms_per_step'], r['launches_per_step']), flush=True)
" | tee -a gpurun_out/r05x/warm_sweep.txt
done; done

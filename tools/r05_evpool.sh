#!/bin/bash
# profiling events from a pool (product library) against created and destroyed per launch (ablation library built from the commit before), interleaved
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05n; mkdir -p $O
run() { timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print('$1: ms/step %.3f  kernel_ms/step %.3f  recall@1 %.3f overflow %d' % (d['ms_per_step'], r['kernel_ms_per_step'], d['recall@1'], d['overflowed_queries']), flush=True)
"; }
for r in 1 2 3 4; do
  run "round $r event pool      "
  LDOT_LIBRARY=$PWD/lightningdot_amd/libldot_ablation.so run "round $r create / destroy"
done | tee $O/ab_event_pool.txt
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; tail -2 $O/pytest_gpu.log

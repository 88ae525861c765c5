#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05p; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_row_shuffle.py tests/test_gpu_search.py -m gpu -q 2>&1 | tail -3 | tee $O/pytest.txt
export LDOT_LIBRARY=$PWD/lightningdot_amd/libldot_ablation.so
for r in 1 2 3; do for nt in 0 1; do
  if [ $nt = 1 ]; then export LDOT_DEBUG_RESCORE_NT=1; else unset LDOT_DEBUG_RESCORE_NT; fi
  timeout 300 python bench.py --steps 15 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print('round $r nt=$nt headline: ms/step %.3f kernel_ms %.3f tail %.3f' % (d['ms_per_step'], r['kernel_ms_per_step'], d['ms_per_step'] - r['kernel_ms_per_step']), flush=True)
" | tee -a $O/ab_rescore_nt.txt
  timeout 300 python bench.py --workload coco --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l)
        print('round $r nt=$nt coco: ms/step %.4f' % d['ms_per_step'], flush=True)
" | tee -a $O/ab_rescore_nt.txt
done; done

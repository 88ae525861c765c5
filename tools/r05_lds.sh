#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05lds; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_search.py tests/test_gpu_configs.py tests/test_gpu_mining.py tests/test_gpu_fuzz.py tests/test_gpu_ivf.py -m gpu -q 2>&1 | tail -3 | tee $O/pytest.txt
for r in 1 2 3; do for lib in libldot_ablation.so libldot.so; do
  for w in flickr coco; do
    LDOT_LIBRARY=$PWD/lightningdot_amd/$lib timeout 300 python bench.py --workload $w --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print('round $r $lib $w: ms/step %.4f  t2i %.4f i2t %.4f kernel_ms %.4f frac %.3f' % (d['ms_per_step'], d['ms_text_to_image'], d['ms_image_to_text'], r['kernel_ms_per_step'], r['frac']), flush=True)
" | tee -a $O/ab_dense_row_stride.txt
  done
  LDOT_LIBRARY=$PWD/lightningdot_amd/$lib timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print('round $r $lib headline: ms/step %.3f kernel_ms %.3f tail %.3f' % (d['ms_per_step'], r['kernel_ms_per_step'], d['ms_per_step'] - r['kernel_ms_per_step']), flush=True)
" | tee -a $O/ab_dense_row_stride.txt
done; done

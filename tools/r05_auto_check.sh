#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r05j}; mkdir -p $O; cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -5 | tee $O/pytest.txt
for w in flickr coco; do timeout 300 python bench.py --workload $w --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print('$w: ms/step %.4f  t2i %.4f i2t %.4f kernel_ms %.4f frac %.3f' % (d['ms_per_step'], d['ms_text_to_image'], d['ms_image_to_text'], r['kernel_ms_per_step'], r['frac']), flush=True)
" | tee -a $O/s2.txt; done

#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_search.py tests/test_gpu_sharded.py tests/test_gpu_configs.py -x -q > $OUT/pytest_prune.txt 2>&1; tail -5 $OUT/pytest_prune.txt
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_prune.json 2> $OUT/bench_prune.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r06/bench_prune.json').read().strip().splitlines()[-1])
s = d['secondary']
print('step %.3f kernel %.3f frac %.3f | flickr %.4f coco %.4f | mining t2i50 %.2f i2t50 %.2f' % (d['ms_per_step'], d['roofline']['kernel_ms_per_step'], d['roofline']['frac'],
      s['flickr_1k']['ms_per_evaluation'], s['coco_5k']['ms_per_evaluation'],
      s['mining_flickr_train']['searches']['t2i_145k_x_29k_top50']['exact_scores']['device_ms'], s['mining_flickr_train']['searches']['i2t_29k_x_145k_top50']['exact_scores']['device_ms']))
PY

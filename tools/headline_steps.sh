for i in 1 2 3 4 5 6; do python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith(chr(123)):
        d = json.loads(l); s = d['secondary']; f = s['flickr_1k']; c = s['coco_5k']
        print('ms/step %.3f median %.3f worst %.3f | flickr median %.4f mean %.4f worst %.3f | coco median %.4f worst %.3f | serving' % (d['ms_per_step'], d['step_ms']['median'], d['step_ms']['worst'], f['ms_per_evaluation'], f['ms_mean'], f['ms_worst'], c['ms_per_evaluation'], c['ms_worst']), {k: round(v['ms'], 4) for k, v in s['serving_latency'].items()})
"; done > gpurun_out/headline_steps.txt 2>&1; cat gpurun_out/headline_steps.txt; timeout 200 python -m pytest tests/test_gpu_bench_contract.py -x -q -m gpu 2>&1 | tail -2

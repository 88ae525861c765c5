#!/bin/bash
# final evidence of round 6: tests, smoke, headline bench (+ CPU baseline + secondary), kernel stats, timeline, S2 lines
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r06final}; mkdir -p $O; cd $GRAFT_REPO_ROOT
timeout 1800 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; tail -3 $O/pytest_gpu.log
timeout 300 python -c 'import __graft_entry__ as g; g.smoke(); print("smoke ok")' > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
timeout 600 python bench.py --workload flickr --steps 50 --warmup 5 > $O/bench_flickr.json 2>> $O/bench.err
timeout 600 python bench.py --workload coco --steps 20 --warmup 3 > $O/bench_coco.json 2>> $O/bench.err
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --force-sharded --backend nccl > $O/bench_sharded_world1_rccl.json 2>> $O/bench.err
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/kt
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt -o k --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 2 --no-cpu-baseline --no-secondary > $O/bench_profiled.json 2> $O/rocprof.err
cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
cd $GRAFT_REPO_ROOT
bash tools/timeline.sh > /dev/null 2>&1; cp gpurun_out/timeline.txt $O/timeline.txt
python - <<'PY'
import json, os
O = os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/r06final/'
d = json.loads([l for l in open(O + 'bench.json') if l.startswith('{')][0]); s = d['secondary']
print('headline', d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_step'], d['roofline']['frac'])
print('secondary flickr %.3f (dev out %.3f) coco %.3f (dev out %.3f) loss %.0f us' % (s['flickr_1k']['ms_per_evaluation'], s['flickr_1k']['ms_per_evaluation_device_outputs'], s['coco_5k']['ms_per_evaluation'], s['coco_5k']['ms_per_evaluation_device_outputs'], s['loss_step']['512x512']['end_to_end_us']))
print('sorted clusters', [(round(r['ms'], 2), r['thresholds'], r['scan_order'], r['redone_queries']) for r in s['sorted_clusters_1m'].get('searches', [])], s['sorted_clusters_1m'].get('error'))
m = s['mining_flickr_train']['searches']
print('mining ids-only ms', {k: round(v['ids_only']['device_ms'], 2) for k, v in m.items() if 'ids_only' in v}, 'whole call s', s['mining_flickr_train'].get('sampled_hard_negatives_wall_s'))
PY

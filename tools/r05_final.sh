#!/bin/bash
# final evidence of the round (after the profiling-event fix): tests, smoke, headline bench (+ CPU baseline + secondary), kernel stats, timeline, S2 lines
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r05final}; mkdir -p $O; cd $GRAFT_REPO_ROOT
timeout 1800 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; tail -3 $O/pytest_gpu.log
timeout 300 python -c 'import __graft_entry__ as g; g.smoke(); print("smoke ok")' > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
timeout 600 python bench.py --workload flickr --steps 50 --warmup 5 > $O/bench_flickr.json 2>> $O/bench.err
timeout 600 python bench.py --workload coco --steps 20 --warmup 3 > $O/bench_coco.json 2>> $O/bench.err
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --force-sharded --backend nccl > $O/bench_sharded_world1_rccl.json 2>> $O/bench.err
timeout 300 tools/bin/mfma_ceiling 12 t16b > $O/mfma_ceiling_t16b.txt 2>&1
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/kt
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt -o k --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 2 --no-cpu-baseline --no-secondary > $O/bench_profiled.json 2> $O/rocprof.err
cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
cd $GRAFT_REPO_ROOT
bash tools/timeline.sh > /dev/null 2>&1; cp gpurun_out/timeline.txt $O/timeline.txt
python - <<'PY'
import json, os
O = os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/r05final/'
d = json.loads([l for l in open(O + 'bench.json') if l.startswith('{')][0]); s = d['secondary']
print('headline', d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_step'], d['roofline']['frac'])
print('secondary flickr %.3f coco %.3f loss %.0f us' % (s['flickr_1k']['ms_per_evaluation'], s['coco_5k']['ms_per_evaluation'], s['loss_step']['512x512']['end_to_end_us']))
PY

#!/bin/bash
# round 5: full GPU test run + headline bench + sharded lines with per-phase times (world 1 on RCCL, 2 / 4 ranks on one GPU over gloo)
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r05e}; mkdir -p $O; cd $GRAFT_REPO_ROOT
timeout 1800 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -5 $O/pytest_gpu.log
timeout 300 python -c 'import __graft_entry__ as g; g.smoke(); print("smoke ok")' > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; cut -c1-300 $O/bench.json
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --force-sharded --backend nccl > $O/bench_sharded_world1_rccl.json 2>> $O/bench.err
timeout 600 python bench.py --gpus 2 --all-on-device0 --backend gloo --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > $O/bench_2ranks_gloo_one_gpu.json 2>> $O/bench.err
timeout 600 python bench.py --gpus 4 --all-on-device0 --backend gloo --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > $O/bench_4ranks_gloo_one_gpu.json 2>> $O/bench.err
python - <<'PY'
import json, os
O = os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/r05e/'
for f in ('bench_sharded_world1_rccl.json', 'bench_2ranks_gloo_one_gpu.json', 'bench_4ranks_gloo_one_gpu.json'):
    try:
        d = json.loads([l for l in open(O + f) if l.startswith('{')][0])
        print(f, 'ranks', d['ranks_seen'], 'ms/step %.3f' % d['ms_per_step'], {k: round(v, 3) for k, v in d['phases_ms_per_step'].items()})
    except Exception as e:
        print(f, 'failed', e)
PY

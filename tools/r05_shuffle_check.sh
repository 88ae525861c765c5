#!/bin/bash
# round 5: row shuffle + regime tests, C client, search / ivf / sharded / contract tests
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r05d}; mkdir -p $O; cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_row_shuffle.py tests/test_gpu_c_abi.py tests/test_gpu_bench_contract.py tests/test_gpu_search.py tests/test_gpu_ivf.py tests/test_gpu_sharded.py -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -40 $O/pytest.log
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --force-sharded --backend nccl 2>/dev/null | tee $O/bench_sharded_world1_rccl.json | cut -c1-200
python - <<'PY'
import json,os
d=json.loads([l for l in open(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r05d/bench_sharded_world1_rccl.json') if l.startswith('{')][0])
print(d['ms_per_step'], d['phases_ms_per_step'])
PY

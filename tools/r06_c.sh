#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06; mkdir -p $OUT
timeout 600 python tools/prune_ab.py 2>&1 | grep -v amdgpu.ids > $OUT/prune_ab.txt; cat $OUT/prune_ab.txt
timeout 1200 python -m pytest tests/test_gpu_mining.py tests/test_gpu_search.py tests/test_gpu_fuzz.py -x -q > $OUT/pytest_big.txt 2>&1; tail -15 $OUT/pytest_big.txt
timeout 600 python tools/mining_bench.py > $OUT/mining_bench_big.json 2> $OUT/mining_bench_big.err; tail -c 3000 $OUT/mining_bench_big.json

#!/bin/bash
# round 6, first GPU call: the oracle-checked mining tests, the mining bench (both modes) and a host profile of the whole call
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_gpu_mining.py -x -q > gpurun_out/r06/mining_tests.txt 2>&1
tail -5 gpurun_out/r06/mining_tests.txt
timeout 600 python tools/mining_bench.py > gpurun_out/r06/mining_bench.json 2> gpurun_out/r06/mining_bench.err
tail -3 gpurun_out/r06/mining_bench.err
timeout 600 python tools/mining_bench.py --cprofile > gpurun_out/r06/mining_cprofile.txt 2>&1
head -60 gpurun_out/r06/mining_cprofile.txt

#!/bin/bash
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_gpu_mining.py -q > gpurun_out/r06/mining_tests.txt 2>&1
tail -12 gpurun_out/r06/mining_tests.txt
bash tools/r06_mining_prof.sh > gpurun_out/r06/mining_prof.log 2>&1
STALL_PROBE_ROUNDS=8 bash tools/stall_trace.sh > gpurun_out/r06/stall_trace.log 2>&1
tail -60 gpurun_out/r06/stall_trace.log | cut -c1-300
